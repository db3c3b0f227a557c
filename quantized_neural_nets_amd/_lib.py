"""ctypes binding of the C ABI in include/gpfq.h (libgpfq_hip.so, built in-tree by csrc/Makefile).

The library is the product: if it is missing this module raises at import -- there is no fallback.
torch is imported first so that the HIP runtime torch bundles (SONAME libamdhip64.so.7) is the one the
library binds to; streams and device pointers are then shared with torch.
"""
import ctypes
import os
import subprocess
import threading

import torch  # noqa: F401  (must precede loading the library, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPFQ_LIB_OVERRIDE") or os.path.join(_HERE, "libgpfq_hip.so")   # override: diagnostic builds

MODE_MSQ, MODE_SOFT, MODE_HARD, MODE_STOCHASTIC = 0, 1, 2, 3
PLAN_AUTO, PLAN_STREAM, PLAN_RESIDENT, PLAN_COOP, PLAN_STREAM_ROWS = 0, 1, 2, 3, 4

EXPORTS = [
    "gpfq_abi_version", "gpfq_last_error", "gpfq_padded_m", "gpfq_workspace_bytes",
    "gpfq_prepare_columns_f32", "gpfq_quantization_f32", "gpfq_quantize_layer_f32", "gpfq_quantizer_f32",
    "gpfq_row_absmax_f32", "gpfq_describe_plan", "gpfq_quantize_groups_prepared_f32", "gpfq_scratch_bytes",
    "gpfq_read_status", "gpfq_column_norms_f32", "gpfq_gather_patches_f32", "gpfq_last_launch_used_exchange",
    "gpfq_describe_plan_mode", "gpfq_prepare_ws_bytes", "gpfq_prepare_columns_ws_f32", "gpfq_philox_uniform_f32",
    "gpfq_spin_limit_word", "gpfq_coop_launch_api_active", "gpfq_clear_contention",
]


class GpfqError(RuntimeError):
    pass


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the extension (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    cmd = ["make", "-s", "-C", src_dir]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "quantized_neural_nets_amd: %s is missing. Build it with `make -C %s` (hipcc, gfx950) or "
            "__graft_entry__.build(); there is no CPU fallback." % (LIB_PATH, os.path.join(_HERE, "csrc")))
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, i64, u64, f32, i32, sz = c.c_void_p, c.c_int64, c.c_uint64, c.c_float, c.c_int, c.c_size_t
    lib.gpfq_abi_version.restype = i32
    lib.gpfq_last_error.restype = c.c_char_p
    lib.gpfq_padded_m.restype = i64
    lib.gpfq_padded_m.argtypes = [i64]
    lib.gpfq_workspace_bytes.restype = sz
    lib.gpfq_workspace_bytes.argtypes = [i64, i64, i64, i32]
    lib.gpfq_prepare_columns_f32.restype = i32
    lib.gpfq_prepare_columns_f32.argtypes = [vp, i64, vp, i64, i64, i64, vp, vp, vp, i64, vp]
    lib.gpfq_prepare_ws_bytes.restype = sz
    lib.gpfq_prepare_ws_bytes.argtypes = [i64, i64]
    lib.gpfq_prepare_columns_ws_f32.restype = i32
    lib.gpfq_prepare_columns_ws_f32.argtypes = [vp, i64, vp, i64, i64, i64, vp, vp, vp, i64, vp, sz, vp]
    lib.gpfq_quantization_f32.restype = i32
    lib.gpfq_quantization_f32.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp, vp, vp, i64, i64, i64, i64,
                                          f32, i32, i32, f32, u64, u64, vp, i64, i32, vp, i32, vp, sz, vp]
    lib.gpfq_quantize_layer_f32.restype = i32
    lib.gpfq_quantize_layer_f32.argtypes = [vp, vp, i64, vp, i64, i64, i64, i64, i32, f32, i32, i32, f32, u64, u64,
                                            vp, vp, i32, vp, vp, vp, sz, i32, vp]
    lib.gpfq_quantize_groups_prepared_f32.restype = i32
    lib.gpfq_quantize_groups_prepared_f32.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, f32, i32, i32,
                                                      f32, u64, u64, vp, i32, vp, i32, vp, sz, vp]
    lib.gpfq_last_launch_used_exchange.restype = i32
    lib.gpfq_spin_limit_word.restype = c.c_uint
    lib.gpfq_coop_launch_api_active.restype = i32
    lib.gpfq_clear_contention.restype = None
    lib.gpfq_column_norms_f32.restype = i32
    lib.gpfq_column_norms_f32.argtypes = [vp, i64, i64, i64, vp, vp]
    lib.gpfq_gather_patches_f32.restype = i32
    lib.gpfq_gather_patches_f32.argtypes = [vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, vp, i64, vp, i64, vp]
    lib.gpfq_philox_uniform_f32.restype = i32
    lib.gpfq_philox_uniform_f32.argtypes = [u64, u64, u64, i64, vp, vp]
    lib.gpfq_scratch_bytes.restype = sz
    lib.gpfq_read_status.restype = i32
    lib.gpfq_read_status.argtypes = [vp, c.POINTER(c.c_int), vp]
    lib.gpfq_quantizer_f32.restype = i32
    lib.gpfq_quantizer_f32.argtypes = [i32, f32, vp, i64, i32, f32, vp, vp, vp, vp]
    lib.gpfq_row_absmax_f32.restype = i32
    lib.gpfq_row_absmax_f32.argtypes = [vp, i64, i64, i64, vp, vp]
    lib.gpfq_describe_plan.restype = i32
    lib.gpfq_describe_plan.argtypes = [i64, i64, i64, i32, i32, c.c_char_p, sz]
    lib.gpfq_describe_plan_mode.restype = i32
    lib.gpfq_describe_plan_mode.argtypes = [i64, i64, i64, i32, i32, i32, c.c_char_p, sz]
    if lib.gpfq_abi_version() != 3:
        raise ImportError("libgpfq_hip.so ABI version mismatch")
    return lib


lib = _load()


from ._digest import kernel_source_digest  # noqa: E402,F401


def check(rc):
    if rc != 0:
        raise GpfqError("gpfq error %d: %s" % (rc, lib.gpfq_last_error().decode()))


def require_gpu_tensor(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GpfqError("%s must be a tensor on the MI355X (cuda) device; this package has no CPU path" % name)
    if t.dtype != torch.float32:
        raise GpfqError("%s must be float32" % name)


def current_stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_scratch = {}           # device index -> scratch buffer
_scratch_user = {}      # device index -> the stream whose launches used the scratch last
_scratch_lock = threading.Lock()
_exclusive = {}         # device index -> re-entrant lock held from scratch() through the launch and its status read


def exclusive(device):
    """The per-device lock a caller holds from `scratch()` through its launches AND the read of their status word.
    The scratch area (exchange granules, status words) is one per device: two host threads on two streams that both
    passed `scratch()` before either launched would run two cooperative grids on the same granules at once, and either
    thread's status read would consume (and clear) a timeout raised by the other's launch.  Inside the lock the order is
    total: the previous user's work is waited for on the device (`scratch()`), this thread's launches are queued, and
    its status read -- a stream synchronisation -- ends before the next thread may even ask for the scratch.
    (`check_status=False` callers release the lock with their launches only QUEUED: the device-side wait of the next
    `scratch()` still keeps the grids apart, but whose timeout a deferred status read reports is then the caller's
    business -- defer only from one thread.)"""
    dev = torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    with _scratch_lock:
        lk = _exclusive.get(index)
        if lk is None:
            lk = _exclusive[index] = threading.RLock()
    return lk


def scratch(device):
    """Scratch area of the cooperative plans (exchange granules + status words), ONE PER DEVICE, allocated and zeroed
    once.  A cooperative grid is sized to be co-resident on an otherwise idle chip (a plain launch behind an occupancy
    query), so two of them running at once on two streams would each be only partly resident and spin to their
    bounds: launches that use the scratch are therefore serialised per device -- a caller on another stream than the
    previous user's first waits (on the device, not the host) for everything that stream has queued.  Call it, launch
    and read the status inside `with exclusive(device):` (host threads; see there)."""
    dev = torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(index)
    with _scratch_lock:
        buf = _scratch.get(index)
        if buf is None:
            buf = torch.zeros((lib.gpfq_scratch_bytes(),), dtype=torch.uint8, device=torch.device("cuda", index))
            _scratch[index] = buf
        last = _scratch_user.get(index)
        if last is not None and last.cuda_stream != cur.cuda_stream:
            cur.wait_stream(last)
        _scratch_user[index] = cur
    return buf


def _read_status(device):
    buf = scratch(device)
    st = (ctypes.c_int * 4)()
    rc = lib.gpfq_read_status(ctypes.c_void_p(buf.data_ptr()), st, current_stream_ptr(buf.device))
    return rc, st


def status_ok(device):
    """Synchronise the current stream and report whether every cooperative launch on it since the last read ran to
    completion (False: one gave up waiting for a peer workgroup; its outputs are invalid).  Clears the status."""
    rc, st = _read_status(device)
    if rc not in (0, -5):
        raise GpfqError("gpfq error %d: %s" % (rc, lib.gpfq_last_error().decode()))
    return rc == 0


def check_status(device):
    """Synchronise and raise if a cooperative kernel reported a timeout since the last check."""
    rc, st = _read_status(device)
    if rc != 0:
        raise GpfqError("gpfq error %d: %s (column %d, row tile %d, member %d)" % (
            rc, lib.gpfq_last_error().decode(), st[1], st[2], st[3]))


def describe_plan(N, d_g, m, groups=1, plan=PLAN_AUTO, mode=MODE_MSQ):
    """One-line description of the plan that launches for this shape and quantizer (include/gpfq.h)."""
    buf = ctypes.create_string_buffer(256)
    rc = lib.gpfq_describe_plan_mode(N, d_g, m, groups, plan, mode, buf, 256)
    if rc < 0:
        check(rc)
    return buf.value.decode()
