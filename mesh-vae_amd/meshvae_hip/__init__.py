"""ctypes binding of libmeshvae_hip.so -- the C ABI declared in include/meshvae_hip.h.

There is no CPU fallback: every op in this package launches hand-written HIP kernels on
an MI355X.  If the shared library is missing, or a tensor is not a contiguous CUDA(HIP)
tensor, the call raises.
"""
import ctypes
import os

# Importing this package changes NOTHING in the process environment.  Two runtime settings matter to it, both read by the
# HIP runtime when IT initialises (the first hipGetDeviceCount: torch.cuda.is_available() already counts), so both belong to
# whoever starts the process:
#   * GPU_STREAMOPS_CP_WAIT=1 -- needed by the ASYNCHRONOUS LAUNCHER (csrc/launcher.hip), an opt-in of the module path
#     (MESHVAE_ASYNC=1): its hipStreamWaitValue64 waits then run on the command processor; by default the runtime executes
#     such a wait as a SHADER that spins on a compute unit for as long as the job runs (the step's kernels: 0.75 instead of
#     0.49 ms, profiles/r04_async_probe.txt).  The launcher is used only when the variable was PRESET to 1 (this package
#     cannot tell whether a value set later was still in time: torch.cuda.is_initialized() stays False after is_available()).
#   * GPU_MAX_HW_QUEUES -- left at the runtime's default (4): the train step keeps three queues busy (caller's stream + two
#     weight-gradient lanes) and a data-parallel rank's collective stream is the fourth (profiles/r05_fourth_queue.txt).
# MESHVAE_ASYNC: "" / "0" = never use the launcher (default); "1" = use it when GPU_STREAMOPS_CP_WAIT=1 is preset;
# "force" = use it regardless (A/B tooling: measures the spinning-shader case).
ASYNC_MODE = os.environ.get("MESHVAE_ASYNC", "0")
CP_WAIT = os.environ.get("GPU_STREAMOPS_CP_WAIT") == "1"

_HERE = os.path.dirname(os.path.abspath(__file__))
# MESHVAE_LIB lets a benchmark A/B two builds of the library in one process-per-run session
LIB_PATH = os.environ.get("MESHVAE_LIB") or os.path.join(_HERE, "libmeshvae_hip.so")
_lib = None


class MeshVaeHipError(RuntimeError):
    pass


class MeshVaeHipUnsupported(MeshVaeHipError):
    """MVH_ERR_UNSUPPORTED: the call is valid but no kernel takes this shape / layout (nothing was written)."""


class CsrStruct(ctypes.Structure):
    """mvh_csr_t"""
    _fields_ = [("n_rows", ctypes.c_int32), ("n_cols", ctypes.c_int32), ("nnz", ctypes.c_int32),
                ("rowptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("val", ctypes.c_void_p),
                ("rowinfo", ctypes.c_void_p), ("ell", ctypes.c_void_p), ("ell_pairs", ctypes.c_int32),
                ("max_row_nnz", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("n_active", ctypes.c_int32), ("sub", ctypes.c_void_p), ("sel_inv", ctypes.c_void_p),
                ("patch", ctypes.c_void_p)]


class PatchPlanStruct(ctypes.Structure):
    """mvh_patch_plan_t (built by meshvae_hip/patches.py)"""
    _fields_ = [("n_patches", ctypes.c_int32), ("n_rings", ctypes.c_int32), ("n_vertices", ctypes.c_int32),
                ("max_rows", ctypes.c_int32), ("max_core", ctypes.c_int32), ("max_excl", ctypes.c_int32),
                ("n_pool_rows", ctypes.c_int32), ("min_core", ctypes.c_int32),
                ("poff", ctypes.c_void_p), ("cnt", ctypes.c_void_p), ("pinfo", ctypes.c_void_p), ("ell", ctypes.c_void_p),
                ("prow_off", ctypes.c_void_p), ("prow_gid", ctypes.c_void_p), ("prow_ptr", ctypes.c_void_p),
                ("pcol", ctypes.c_void_p), ("pval", ctypes.c_void_p), ("pool_rowptr", ctypes.c_void_p),
                ("max_pool_nnz", ctypes.c_int32), ("u_rows", ctypes.c_int32), ("urec", ctypes.c_void_p)]


VAE_MAX_LAYERS = 8


class VaeDesc(ctypes.Structure):
    """mvh_vae_desc_t"""
    _fields_ = [("n_layers", ctypes.c_int32), ("num_features", ctypes.c_int32), ("num_hidden", ctypes.c_int32),
                ("num_classes", ctypes.c_int32), ("num_style", ctypes.c_int32), ("dropout_p", ctypes.c_float),
                ("storage", ctypes.c_int32),
                ("filters", ctypes.c_int32 * (VAE_MAX_LAYERS + 2)), ("K", ctypes.c_int32 * (VAE_MAX_LAYERS + 1)),
                ("num_nodes", ctypes.c_int32 * (VAE_MAX_LAYERS + 1)),
                ("lap", CsrStruct * (VAE_MAX_LAYERS + 1)), ("lap_t", CsrStruct * (VAE_MAX_LAYERS + 1)),
                ("down", CsrStruct * VAE_MAX_LAYERS), ("down_t", CsrStruct * VAE_MAX_LAYERS),
                ("up", CsrStruct * VAE_MAX_LAYERS), ("up_t", CsrStruct * VAE_MAX_LAYERS)]


CSR_NORMALIZED_LAPLACIAN, CSR_SYMMETRIC, CSR_SELECTION, CSR_ELL_OVERFLOW = 1, 2, 4, 8
STORAGE_F32, STORAGE_BF16 = 0, 1
ABI_VERSION = 321   # MVH_ABI_VERSION of include/meshvae_hip.h this binding was written against
_P, _I, _F, _Z = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_size_t
_CSR = ctypes.POINTER(CsrStruct)

# name -> (restype, argtypes); must list EVERY symbol of include/meshvae_hip.h (tests check it)
SIGNATURES = {
    "mvh_version": (ctypes.c_int, []),
    "mvh_last_error": (ctypes.c_char_p, []),
    "mvh_device_info": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_int]),
    "mvh_debug_set": (ctypes.c_int, [ctypes.c_char_p, _I]),
    "mvh_debug_get": (_I, [ctypes.c_char_p]),
    "mvh_spmm": (ctypes.c_int, [_P, _CSR, _P, _P, _P, _P, _F, _F, _I, _I, _I]),
    "mvh_pool_fwd": (ctypes.c_int, [_P, _CSR, _P, _P, _I, _I]),
    "mvh_pool_bwd": (ctypes.c_int, [_P, _CSR, _P, _P, _I, _I]),
    "mvh_cheb_conv_ws_bytes": (_Z, [_I] * 5),
    "mvh_cheb_conv_fwd": (ctypes.c_int, [_P, _CSR, _P, _P, _P, _P, _P] + [_I] * 6 + [_P, _Z]),
    "mvh_cheb_conv_bwd_ws_bytes": (_Z, [_I] * 5),
    "mvh_cheb_conv_bwd": (ctypes.c_int, [_P, _CSR, _CSR] + [_P] * 8 + [_I] * 6 + [_P, _Z]),
    "mvh_cheb_conv_fwd_signs": (ctypes.c_int, [_P, _CSR, _P, _P, _P, _P, _P] + [_I] * 5 + [_P, _Z]),
    "mvh_cheb_conv_bwd_signs": (ctypes.c_int, [_P, _CSR, _CSR] + [_P] * 8 + [_I] * 5 + [_P, _Z]),
    "mvh_cheb_conv_fwd_bf16": (ctypes.c_int, [_P, _CSR, _P, _P, _P, _P, _P] + [_I] * 6 + [_P, _Z]),
    "mvh_cheb_conv_bwd_bf16": (ctypes.c_int, [_P, _CSR, _CSR] + [_P] * 7 + [_I] * 6 + [_P, _Z]),
    "mvh_cheb_conv_strided_ws_bytes": (_Z, [_I] * 5),
    "mvh_cheb_conv_fwd_strided": (ctypes.c_int, [_P, _CSR, _P, ctypes.c_int64, ctypes.c_int64, _P, _P, _P, _P] + [_I] * 6 + [_P, _Z]),
    "mvh_cheb_conv_bwd_strided": (ctypes.c_int, [_P, _CSR, _CSR, _P, ctypes.c_int64, ctypes.c_int64] + [_P] * 7 + [_I] * 6 + [_P, _Z]),
    "mvh_pool_fwd_strided": (ctypes.c_int, [_P, _CSR, _P, ctypes.c_int64, ctypes.c_int64, _P, _I, _I]),
    "mvh_linear_fwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _F]),
    "mvh_linear_bwd": (ctypes.c_int, [_P] * 8 + [_I] * 4 + [_F, _P, _Z]),
    "mvh_vae_latent_fwd": (ctypes.c_int, [_P, _P, _P, _P, _F] + [_P] * 12 + [_I] * 4),
    "mvh_vae_latent_bwd": (ctypes.c_int, [_P, _P, _P, _P, _F] + [_P] * 17 + [_I] * 4 + [_P, _Z]),
    "mvh_vae_loss_ws_bytes": (_Z, [_I]),
    "mvh_vae_loss_fwd": (ctypes.c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P] + [_I] * 4 + [_P, _Z]),
    "mvh_recon_postprocess": (ctypes.c_int, [_P] * 10 + [_I, _I]),
    "mvh_vae_wait_dense_grads": (ctypes.c_int, [_P]),
    "mvh_vae_encode": (ctypes.c_int, [_P, ctypes.POINTER(VaeDesc), _P, _P, _P, _I, _P, _P, _Z]),
    "mvh_vae_decode": (ctypes.c_int, [_P, ctypes.POINTER(VaeDesc), _P, _P, _P, _I, _P, _P, _Z]),
    "mvh_procrustes_stats": (ctypes.c_int, [_P] * 4 + [_I, _I]),
    "mvh_procrustes_apply": (ctypes.c_int, [_P] * 8 + [_I, _I]),
    "mvh_gather_normalize": (ctypes.c_int, [_P, _P, ctypes.c_int64, _P, _P, _P, _P, _P, _I, ctypes.c_int64]),
    "mvh_adam_step": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int64, _F, _F, _F, _F, _F, _F, _P,
                                      ctypes.c_int64, ctypes.c_int64]),
    "mvh_adam_step_counted": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int64, _F, _F, _F, _F, _F, _F, _P, _I,
                                              ctypes.c_int64, ctypes.c_int64]),
    "mvh_sizeof_vae_desc": (_Z, []),
    "mvh_sizeof_csr": (_Z, []),
    "mvh_vae_step_ws_bytes": (_Z, [ctypes.POINTER(VaeDesc), _I]),
    "mvh_vae_param_count": (ctypes.c_int32, [ctypes.POINTER(VaeDesc)]),
    "mvh_vae_ws_offset": (ctypes.c_int64, [ctypes.POINTER(VaeDesc), _I, ctypes.c_char_p, _I, ctypes.POINTER(ctypes.c_int64)]),
    "mvh_vae_forward": (ctypes.c_int, [_P, ctypes.POINTER(VaeDesc), _P, _P, _P, _P, _I, _P, _P, _I, _F] + [_P] * 9 + [_P, _Z]),
    "mvh_vae_backward": (ctypes.c_int, [_P, ctypes.POINTER(VaeDesc), _P, _P, _P, _P, _P, _I, _P, _P, _I, _F] + [_P] * 5 + [_P, _Z, _P]),
    "mvh_vae_backward_prefetch": (ctypes.c_int, [_P, ctypes.POINTER(VaeDesc), _P, _I, _P, _Z, _P]),
    "mvh_launcher_supported": (ctypes.c_int, []),
    "mvh_launcher_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "mvh_launcher_sync": (ctypes.c_int, [_P]),
    "mvh_launcher_destroy": (ctypes.c_int, [_P]),
    "mvh_launcher_test_job": (ctypes.c_int, [_P, _P, _I]),
    "mvh_vae_forward_async": (ctypes.c_int, [_P, _P, ctypes.POINTER(VaeDesc), _P, _P, _P, _P, _I, _P, _P, _I, _F] + [_P] * 9 + [_P, _Z]),
    "mvh_vae_backward_async": (ctypes.c_int, [_P, _P, ctypes.POINTER(VaeDesc), _P, _P, _P, _P, _P, _I, _P, _P, _I, _F] + [_P] * 5 + [_P, _Z]),
    "mvh_vae_loss_bwd": (ctypes.c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P] + [_I] * 4),
}


def lib():
    """Load libmeshvae_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MeshVaeHipError(
                f"{LIB_PATH} is missing: build it with `make -C mesh-vae_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        # torch first: its wheel bundles its own libamdhip64; if this library were loaded before torch, the loader would
        # bind it to /opt/rocm's copy and the process would hold two HIP runtimes (the second one sees no device)
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.mvh_version() != ABI_VERSION:
            raise MeshVaeHipError(f"libmeshvae_hip.so speaks ABI {handle.mvh_version()}, this binding {ABI_VERSION} (stale build?)")
        if handle.mvh_sizeof_csr() != ctypes.sizeof(CsrStruct) or handle.mvh_sizeof_vae_desc() != ctypes.sizeof(VaeDesc):
            raise MeshVaeHipError("ctypes struct layout does not match libmeshvae_hip.so (stale build?)")
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        cls = MeshVaeHipUnsupported if rc == 3 else MeshVaeHipError
        raise cls(f"libmeshvae_hip error {rc}: {lib().mvh_last_error().decode()}")


_launchers = {}
_launcher_pid = None


def launcher(device_index):
    """The asynchronous launcher of a device (csrc/launcher.hip; one per process and device, created on first use and
    destroyed at interpreter exit), or None when the device has no hipStreamWaitValue64 / MESHVAE_ASYNC=0."""
    if device_index in _launchers:
        return _launchers[device_index]
    handle = None
    mode = ASYNC_MODE                                       # opt-in: see the top of this file
    if (mode == "1" and CP_WAIT and os.environ.get("WORLD_SIZE", "1") in ("", "1")) or mode == "force":
        import torch
        with torch.cuda.device(device_index):
            if lib().mvh_launcher_supported():
                h = ctypes.c_void_p()
                check(lib().mvh_launcher_create(ctypes.byref(h)))
                handle = h
                global _launcher_pid
                if _launcher_pid is None:
                    import atexit
                    _launcher_pid = os.getpid()
                    atexit.register(_destroy_launchers)
    _launchers[device_index] = handle
    return handle


def _destroy_launchers():
    if os.getpid() != _launcher_pid:        # a forked child (a DataLoader worker) owns neither the thread nor the stream
        return
    for k, h in list(_launchers.items()):
        if h is not None:
            # a job that failed AFTER the last launcher call of the script has told nobody yet: say so now (its outputs --
            # a last eval batch, a single inference call -- are uninitialised memory)
            if lib().mvh_launcher_sync(h) != 0:
                import sys
                print(f"meshvae_hip: an asynchronous job on device {k} failed and was never reported: "
                      f"{lib().mvh_last_error().decode()}", file=sys.stderr)
            lib().mvh_launcher_destroy(h)
        _launchers[k] = None


class debug_switch:
    """Context manager: set one of the library's debug switches (include/meshvae_hip.h, mvh_debug_set) and
    restore it afterwards.  Test / A-B tooling; the product never calls it."""

    def __init__(self, key, value):
        self.key, self.value = key.encode(), int(value)

    def __enter__(self):
        self.old = lib().mvh_debug_get(self.key)
        check(lib().mvh_debug_set(self.key, self.value))
        return self

    def __exit__(self, *exc):
        check(lib().mvh_debug_set(self.key, self.old))


def device_info():
    n_cu, lds = ctypes.c_int(0), ctypes.c_int(0)
    arch = ctypes.create_string_buffer(64)
    check(lib().mvh_device_info(ctypes.byref(n_cu), ctypes.byref(lds), arch, 64))
    return {"n_cu": n_cu.value, "lds_bytes_per_cu": lds.value, "arch": arch.value.decode()}
