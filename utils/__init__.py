"""Import shims: `metrics` and `mm3d_pn2` resolve to mvp_benchmark_amd."""
