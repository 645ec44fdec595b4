from . import reparam, sdf_direct_reparam, sdf_silhouette_reparam, sdf_simple_shading_reparam  # noqa: F401  (plugin registration)
