"""ieee_amd — MI355X (gfx950) native implementation of the IEEE3modalPart hot
path of ziwang1121/IEEE: hand-written HIP kernels behind a C ABI
(include/ieee_amd.h, ieee_amd/libieee_amd.so) plus a thin host-side mirror of
the reference's Python surface (build_model / Image3MEngine /
compute_distance_matrix / evaluate_rank).  There is no CPU fallback: every
compute entry point raises if the HIP library or a gfx950 GPU is missing."""

import os as _os

# Hardware queues per stream priority (ROCclr's GPU_MAX_HW_QUEUES, default 4; read when the HIP runtime initialises, i.e. at
# the first HIP call of the process -- import ieee_amd before touching torch.cuda).  The train step runs on 4-6 streams
# (compute, weight gradients at low priority, branch, optimizer / communication, torch's high-priority collective stream);
# with 4 or more queues per priority, and the collective stream created BEFORE the executor's streams -- the order of a
# torchrun job -- the two main streams of the step stop overlapping on this runtime: 22.9 ms per step instead of 14.7
# (8 and 16: 20.5-21 ms in either order; 1, 2 and 3: 14.5-15.1 in every order tried; scripts/dp_order_probe.py, LABNOTES.md
# section 6).  In a data-parallel job (WORLD_SIZE > 1) ONE queue per priority: the compute stream then never shares a queue
# with the communication stream (which runs at high priority there, beside torch's collective stream), so the gradient
# all-reduces overlap the backward whatever order the streams were created in -- with 2 queues they overlap only when the
# executor's streams exist before the process group's.  An explicit setting of the caller wins.
def _world_size():
    try:
        return int(_os.environ.get("WORLD_SIZE", "1") or 1)
    except ValueError:
        return 1


def _decide_hw_queues():
    """-> the queue count the HIP runtime of this process runs (or will start) with.  The variable is written only when the
    runtime has not started yet: a value exported after that would describe a configuration the process is not in (the engine
    picks the communication stream's priority from HW_QUEUES, not from the environment)."""
    explicit = _os.environ.get("GPU_MAX_HW_QUEUES")
    if explicit is not None:
        try:
            return int(explicit)
        except ValueError:
            return 4
    import sys as _sys
    _torch = _sys.modules.get("torch")
    if _torch is not None and getattr(_torch, "cuda", None) is not None and _torch.cuda.is_initialized():
        import warnings as _warnings
        _warnings.warn("ieee_amd: the HIP runtime of this process started before `import ieee_amd` could set GPU_MAX_HW_QUEUES; with the "
                       "runtime's default of 4 hardware queues per stream priority the train step's streams may not overlap (22.9 instead "
                       "of 14.7 ms per step in a data-parallel job). Export GPU_MAX_HW_QUEUES=2 (1 in a multi-process job) or import "
                       "ieee_amd first.")
        return 4
    n = 1 if _world_size() > 1 else 2
    _os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return n


HW_QUEUES = _decide_hw_queues()

__version__ = "0.1.0"
