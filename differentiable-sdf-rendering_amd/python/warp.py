"""Warp fields of the path (python/warp.py).  In the reference these objects evaluate the
reparameterisation with Dr.Jit; here they only carry its configuration -- the evaluation is
inside the render kernels (csrc/dsdf_math.h: warp_coefficients) -- and select it."""


class WarpField2D:
    """Normal-aligned warp field with analytic divergence (python/warp.py:7-128)."""
    reparameterize = True

    def __init__(self, sdf, weight_strategy=4, edge_eps=0.05):
        self.sdf = sdf
        self.max_reparam_depth = -1
        self.edge_eps = edge_eps / 4 if weight_strategy == 2 else edge_eps      # python/warp.py:22-23
        self.weight_strategy = weight_strategy
        self.clamping_thresh = 0.0
        self.return_aovs = False
        self.normalize_warp_field = True

    def apply(self, params):
        """Writes this field's settings into a DsdfParams struct."""
        # (return_aovs, python/warp.py:17, 105-106, is read by ReparamIntegrator.render: the debug channels are a render of
        # their own, dsdf_render_aovs, and change nothing in the parameters of the passes)
        params.edge_eps = float(self.edge_eps)
        params.weight_strategy = int(self.weight_strategy)
        params.clamping_thresh = float(self.clamping_thresh)
        params.normalize_warp_field = int(bool(self.normalize_warp_field))          # python/warp.py:56-62
        params.max_reparam_depth = int(self.max_reparam_depth)                      # python/warp.py:103
        return params


class DummyWarpField:
    """No reparameterisation (python/warp.py:179-196): shading gradients only."""
    reparameterize = False

    def __init__(self, sdf):
        self.sdf = sdf
        self.return_aovs = False

    def apply(self, params):
        return params


class WarpFieldConvolution:
    """Bangaru et al. 2020 baseline (python/warp.py:131-176): comparison method, out of scope."""

    def __init__(self, sdf, n_aux_rays=16):
        raise NotImplementedError("WarpFieldConvolution is a paper baseline outside the supported path (DESIGN.md section 9)")
