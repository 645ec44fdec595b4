from _scene import mis_weight                                                         # noqa: F401
