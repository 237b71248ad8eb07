"""ieee_amd — MI355X (gfx950) native implementation of the IEEE3modalPart hot
path of ziwang1121/IEEE: hand-written HIP kernels behind a C ABI
(include/ieee_amd.h, ieee_amd/libieee_amd.so) plus a thin host-side mirror of
the reference's Python surface (build_model / Image3MEngine /
compute_distance_matrix / evaluate_rank).  There is no CPU fallback: every
compute entry point raises if the HIP library or a gfx950 GPU is missing."""

__version__ = "0.1.0"
