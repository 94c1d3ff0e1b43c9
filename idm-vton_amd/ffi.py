"""ctypes binding of libidmvton_hip.so -- a field-for-field mirror of include/idmvton_hip.h.

The mirror is verified at load time against `idmvton_sizeof()` exported by the library, so a drift between this file
and the header fails loudly instead of corrupting kernel arguments.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IDMVTON_HIP_LIB") or os.path.join(_HERE, "libidmvton_hip.so")   # override: A/B of library builds

F16, BF16, F32, F8E4M3 = 0, 1, 2, 3
EPI_NONE, EPI_GEGLU, EPI_GELU, EPI_QUICKGELU, EPI_XATTN = 0, 1, 2, 3, 4
ATTN_SELF, ATTN_CROSS = 0, 1
IO_RES_F32, IO_OUT_F32, IO_BIAS_F32, IO_OUT_F8 = 1, 2, 4, 8
GN_X_F32, GN_Y_SPLIT, GN_AFFINE_F32 = 1, 2, 4
LAYOUT_SPLIT, LAYOUT_NHWC_F32 = 1, 2
SPLIT_ACT, SPLIT_W3, SPLIT_W3T = 0, 1, 2
MAX_SEG = 24
ABI_VERSION = 9

i32, u32, f32, vp = C.c_int32, C.c_uint32, C.c_float, C.c_void_p


class Seg(C.Structure):
    _fields_ = [("ptr", vp), ("bytes", u32), ("pitch", i32), ("coff", i32), ("len", i32), ("dy", i32), ("dx", i32)]


class XAttn(C.Structure):
    _fields_ = [("nseg", i32), ("k", vp * 2), ("ldk", i32 * 2), ("k_rows", i32 * 2), ("vt", vp * 2), ("ldvt", i32 * 2), ("nk", i32 * 2),
                ("tokens", i32), ("ip_scale", f32)]


class GemmConvArgs(C.Structure):
    _fields_ = [("dtype", i32), ("w", vp), ("N", i32), ("Ktot", i32), ("nseg", i32), ("seg", Seg * MAX_SEG),
                ("M", i32), ("Ho", i32), ("Wo", i32), ("Hi", i32), ("Wi", i32), ("stride", i32), ("ups", i32),
                ("out", vp), ("ldo", i32), ("bias", vp), ("rowbias", vp), ("rowbias_ld", i32),
                ("rows_per_group", i32), ("res", vp), ("ldr", i32), ("mode", i32), ("vt", vp), ("vt_n0", i32),
                ("vt_tokens", i32), ("tile_hint", i32), ("vt_perm", i32), ("io_flags", i32), ("xattn", C.POINTER(XAttn)),
                ("colscale_n", i32), ("colscale", f32), ("f8_out_scale", f32), ("f8_vt_scale", f32)]


class AttnArgs(C.Structure):
    _fields_ = [("dtype", i32), ("mode", i32), ("B", i32), ("heads", i32), ("Nq", i32), ("q", vp), ("ldq", i32),
                ("out", vp), ("ldo", i32), ("nseg", i32), ("k", vp * 2), ("ldk", i32 * 2), ("vt", vp * 2),
                ("ldvt", i32 * 2), ("nk", i32 * 2), ("k_rows", i32 * 2), ("seg_b0", i32 * 2), ("ip_scale", f32), ("tune", i32), ("q_prescaled", i32)]


class LayerNormArgs(C.Structure):
    _fields_ = [("dtype", i32), ("rows", i32), ("C", i32), ("x", vp), ("ldx", i32), ("gamma", vp), ("beta", vp),
                ("eps", f32), ("y", vp), ("ldy", i32), ("y2", vp), ("ldy2", i32), ("x_f32", i32)]


class GroupNormArgs(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("HW", i32), ("C", i32), ("groups", i32), ("x", vp), ("C1", i32),
                ("x2", vp), ("gamma", vp), ("beta", vp), ("eps", f32), ("silu", i32), ("y", vp), ("stats", vp), ("stats_doubles", i32),
                ("flags", i32)]


class PackInputArgs(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("hw", i32), ("cpad", i32), ("latents", vp), ("cond", vp), ("out", vp)]


class CfgStepArgs(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("hw", i32), ("ldc", i32), ("eps_nhwc", vp), ("latents", vp),
                ("noise", vp), ("coef", vp)]


class LayoutArgs(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("C", i32), ("HW", i32), ("cpad", i32), ("to_nhwc", i32), ("src", vp),
                ("dst", vp), ("scale", f32), ("shift", f32), ("flags", i32)]


class VaeSampleArgs(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("hw", i32), ("ldm", i32), ("moments", vp), ("noise", vp), ("z", vp),
                ("scale", f32)]


class AttnSmallArgs(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("heads", i32), ("Lq", i32), ("Lk", i32), ("d", i32), ("q", vp), ("ldq", i32),
                ("k", vp), ("ldk", i32), ("v", vp), ("ldv", i32), ("out", vp), ("ldo", i32), ("scale", f32), ("causal", i32)]


class AttnF8Args(C.Structure):
    _fields_ = [("out_dtype", i32), ("B", i32), ("heads", i32), ("Nq", i32), ("q8", vp), ("ldq", i32), ("out", vp), ("ldo", i32),
                ("nseg", i32), ("k8", vp * 2), ("ldk", i32 * 2), ("vt8", vp * 2), ("ldvt", i32 * 2), ("nk", i32 * 2), ("k_rows", i32 * 2),
                ("seg_b0", i32 * 2), ("qk_scale_exp", i32), ("v_scale_exp", i32)]


class QuantF8Args(C.Structure):
    _fields_ = [("dtype", i32), ("mode", i32), ("rows", i32), ("cols", i32), ("src", vp), ("lds", i32), ("dst", vp), ("ldd", i32), ("scale", f32)]


class SoftmaxArgs(C.Structure):
    _fields_ = [("dtype", i32), ("rows", i32), ("n", i32), ("ld", i32), ("x", vp), ("scale", f32), ("y_split", vp), ("ldy", i32), ("n_valid", i32)]


class SplitArgs(C.Structure):
    _fields_ = [("dtype", i32), ("mode", i32), ("rows", i32), ("cols", i32), ("src", vp), ("lds", i32), ("dst", vp), ("ldd", i32)]


STRUCTS = {"idmvton_seg": Seg, "idmvton_gemm_conv_args": GemmConvArgs, "idmvton_attn_args": AttnArgs,
           "idmvton_layernorm_args": LayerNormArgs, "idmvton_groupnorm_args": GroupNormArgs,
           "idmvton_pack_input_args": PackInputArgs, "idmvton_cfg_step_args": CfgStepArgs,
           "idmvton_layout_args": LayoutArgs, "idmvton_vae_sample_args": VaeSampleArgs, "idmvton_softmax_args": SoftmaxArgs,
           "idmvton_attn_small_args": AttnSmallArgs, "idmvton_attn_f8_args": AttnF8Args, "idmvton_quant_f8_args": QuantF8Args,
           "idmvton_split_args": SplitArgs, "idmvton_xattn": XAttn}

# every symbol include/idmvton_hip.h declares
SYMBOLS = ["idmvton_last_error", "idmvton_abi_version", "idmvton_sizeof", "idmvton_gemm_conv", "idmvton_attn_fwd",
           "idmvton_layernorm", "idmvton_groupnorm", "idmvton_pack_input", "idmvton_cfg_step", "idmvton_layout",
           "idmvton_vae_sample", "idmvton_softmax_rows", "idmvton_probe_mfma", "idmvton_groupnorm_stats_doubles",
           "idmvton_attn_small", "idmvton_rccl_unique_id", "idmvton_rccl_comm_init", "idmvton_rccl_bcast_arena",
           "idmvton_rccl_comm_destroy", "idmvton_attn_f8", "idmvton_quant_f8", "idmvton_split"]

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load libidmvton_hip.so (once).  Raises HipLibraryMissing -- never falls back to another implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(f"{LIB_PATH} not built: run `make -C {os.path.join(_HERE, 'csrc')}` "
                                "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(L, s):
            raise HipLibraryMissing(f"{LIB_PATH} does not export {s}")
    L.idmvton_last_error.restype = C.c_char_p
    L.idmvton_sizeof.argtypes = [C.c_char_p]
    L.idmvton_abi_version.restype = C.c_int
    v = L.idmvton_abi_version()
    if v != ABI_VERSION:                                  # semantic changes do not always move a struct size (v3 -> v5 did not): check the number too
        raise HipLibraryMissing(f"{LIB_PATH} implements C ABI version {v}, this binding needs {ABI_VERSION}: rebuild it "
                                f"(`make -C {os.path.join(_HERE, 'csrc')}`)")
    for name, st in STRUCTS.items():
        n = L.idmvton_sizeof(name.encode())
        if n != C.sizeof(st):
            raise HipLibraryMissing(f"ABI drift: sizeof({name}) is {n} in the library, {C.sizeof(st)} in ffi.py")
    for s in ("idmvton_gemm_conv", "idmvton_attn_fwd", "idmvton_layernorm", "idmvton_groupnorm",
              "idmvton_pack_input", "idmvton_cfg_step", "idmvton_layout", "idmvton_vae_sample", "idmvton_softmax_rows",
              "idmvton_attn_small", "idmvton_attn_f8", "idmvton_quant_f8", "idmvton_split"):
        getattr(L, s).argtypes = [vp, vp]
        getattr(L, s).restype = C.c_int
    L.idmvton_probe_mfma.argtypes = [C.c_int, vp, vp, vp, vp]
    L.idmvton_groupnorm_stats_doubles.argtypes = [C.c_int] * 4
    L.idmvton_groupnorm_stats_doubles.restype = C.c_int
    L.idmvton_rccl_unique_id.argtypes = [vp]
    L.idmvton_rccl_comm_init.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.idmvton_rccl_bcast_arena.argtypes = [vp, vp, C.c_uint64, C.c_int, C.c_uint64, vp]
    L.idmvton_rccl_comm_destroy.argtypes = [vp]
    for s in ("idmvton_rccl_unique_id", "idmvton_rccl_comm_init", "idmvton_rccl_bcast_arena", "idmvton_rccl_comm_destroy"):
        getattr(L, s).restype = C.c_int
    _lib = L
    return L


def call(fn_name, args, stream):
    """Invoke an `int f(const args*, void* stream)` entry point; raise RuntimeError with the library's message."""
    L = lib()
    rc = getattr(L, fn_name)(C.byref(args), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"{fn_name} failed ({rc}): {L.idmvton_last_error().decode()}")
