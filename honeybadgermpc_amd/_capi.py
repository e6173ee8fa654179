"""
ctypes binding of ``libhbmpc_hip.so`` (C ABI declared in ``include/hbmpc_hip.h``).

This is the thin FFI layer the north star asks for: host code stays Python, the
arithmetic lives in hand-written HIP for gfx950.  PyTorch is used only as plumbing
(device buffers, current stream); nothing in the C signatures is a torch type.

There is NO CPU fallback.  If the shared library is missing or no MI355X is
visible, every entry point raises ``HbmpcBackendError``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HBMPC_HIP_LIB: load another build of the same library (A/B timing of kernel revisions on one GPU box)
LIB_PATH = os.environ.get("HBMPC_HIP_LIB") or os.path.join(_HERE, "lib", "libhbmpc_hip.so")

HB_OK, HB_ERR_SINGULAR, HB_ERR_BAD_ARG, HB_ERR_UNSUPPORTED = 0, 1, 2, 3
HB_ERR_NO_DEVICE, HB_ERR_HIP, HB_ERR_MISMATCH, HB_ERR_RETRY = 4, 5, 6, 7
HB_DEC_COLLECTING, HB_DEC_DONE, HB_DEC_DISAGREE, HB_DEC_UNSUPPORTED, HB_DEC_PENDING = 0, 1, 2, 3, 4
HB_DEC_OPT_DEFER, HB_DEC_OPT_BESIDE = 1, 2

_STATUS_NAMES = {
    1: "HB_ERR_SINGULAR",
    2: "HB_ERR_BAD_ARG",
    3: "HB_ERR_UNSUPPORTED",
    4: "HB_ERR_NO_DEVICE",
    5: "HB_ERR_HIP",
    6: "HB_ERR_MISMATCH",
    7: "HB_ERR_RETRY",
}


class HbmpcBackendError(RuntimeError):
    """The HIP backend is unavailable or a HIP call failed."""


class HbView(ctypes.Structure):
    _fields_ = [("stride_c", ctypes.c_int64), ("stride_l", ctypes.c_int64)]


# every symbol include/hbmpc_hip.h declares: (restype, argtypes)
_vp, _i, _i64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
_pp = ctypes.POINTER(ctypes.c_void_p)
SYMBOLS = {
    "hb_version": (_i, []),
    "hb_device_count": (_i, []),
    "hb_ctx_create": (_i, [_pp, _vp, _i, _i]),
    "hb_ctx_destroy": (None, [_vp]),
    "hb_last_error": (ctypes.c_char_p, [_vp]),
    "hb_ctx_cache_clear": (_i, [_vp]),
    "hb_ctx_cache_entries": (_i, [_vp]),
    "hb_elem_bytes": (_i, [_vp]),
    "hb_malloc": (_i, [_vp, _pp, _sz]),
    "hb_free": (_i, [_vp, _vp]),
    "hb_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "hb_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "hb_stream_sync": (_i, [_vp, _vp]),
    "hb_vand_matrix_create": (_i, [_vp, _vp, _i, _i, _pp, _vp]),
    "hb_vand_inverse_create": (_i, [_vp, _vp, _i, _pp, _vp]),
    "hb_matrix_from_host": (_i, [_vp, _vp, _i, _i, _pp, _vp]),
    "hb_matrix_to_host": (_i, [_vp, _vp, _vp, _vp]),
    "hb_matrix_destroy": (None, [_vp]),
    "hb_reduce": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "hb_quick_interp_check": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "hb_quick_interp_check_map": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "hb_quick_dec_create": (_i, [_vp, _vp, _i, _pp, _vp]),
    "hb_quick_dec_arrivals": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "hb_quick_dec_decide": (_i, [_vp, _vp, _i, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "hb_quick_dec_launch": (_i, [_vp, _vp, _i, _vp, _i64, _i64, _i64, _vp, _vp]),
    "hb_quick_dec_verdict": (_i, [_vp, _vp, _vp]),
    "hb_quick_dec_beside": (_i, [_vp, _i]),
    "hb_quick_dec_destroy": (None, [_vp]),
    "hb_dec_create": (_i, [_vp, _vp, _i, _i, _i, _pp, _vp]),
    "hb_dec_begin": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _i, _vp]),
    "hb_dec_arrived1": (_i, [_vp, _i]),
    "hb_dec_options": (_i, [_vp, _i]),
    "hb_dec_settle": (_i, [_vp]),
    "hb_dec_arrived": (_i, [_vp, _vp, _i, _vp, _vp]),
    "hb_dec_verdict": (_i, [_vp, _vp, _vp]),
    "hb_dec_arrivals_list": (_i, [_vp, _vp, _i, _vp]),
    "hb_dec_destroy": (None, [_vp]),
    "hb_wait_create": (_i, [_vp, _i, _pp]),
    "hb_wait_begin": (_i, [_vp, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _vp, _vp]),
    "hb_wait_arrived1": (_i, [_vp, _i, _i]),
    "hb_wait_result": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "hb_wait_destroy": (None, [_vp]),
    "hb_symbols_fetch": (_i, [_vp, _vp, _i, _i64, _i64, _vp, _i, _vp, _vp]),
    "hb_candidate_check": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i64, _i64, _vp, _vp, _vp]),
    "hb_stream_after": (_i, [_vp, _vp, _vp]),
    "hb_side_stream": (_i, [_vp, _pp]),
    "hb_probe_create": (_i, [_vp, _vp, _i, _i, _pp, _vp]),
    "hb_probe_feed": (_i, [_vp, _vp, _i, _vp, _i64, _i64, _i, _vp, _vp, _vp]),
    "hb_probe_reset": (_i, [_vp]),
    "hb_probe_workgroups": (_i, [_vp, _i]),
    "hb_probe_points_fed": (_i, [_vp]),
    "hb_probe_destroy": (None, [_vp]),
    "hb_matvec": (_i, [_vp, _vp, _vp, HbView, _vp, _vp, HbView, _i64, _vp]),
    "hb_matvec_check": (_i, [_vp, _vp, _vp, HbView, _vp, _vp, HbView, _vp, _i, _vp, _i64, _vp]),
    "hb_vandermonde_batch_evaluate": (_i, [_vp, _vp, _i, _vp, _i64, _i, _vp, _vp]),
    "hb_vandermonde_batch_interpolate": (_i, [_vp, _vp, _i, _vp, _i64, _vp, _vp]),
    "hb_fft_batch_evaluate": (_i, [_vp, _vp, _i, _vp, _i64, _i, _i, _vp, _vp]),
    "hb_fft_batch_interpolate": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i64, _vp, _vp]),
    "hb_gao_decode": (_i, [_vp, _vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "hb_wb_decode": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "hb_sqrt_mod": (_i, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "hb_open_plan_create": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i64, _pp, _vp]),
    "hb_open_r1_encode": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "hb_open_r1_decode": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "hb_open_r2_decode": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "hb_open_status": (_i, [_vp, _vp]),
    "hb_open_plan_set_option": (_i, [_vp, _i, _i]),
    "hb_open_plan_get_option": (_i, [_vp, _i, _vp]),
    "hb_open_plan_destroy": (None, [_vp]),
    "hb_selftest_mulmod": (_i, [_vp, _i, _vp, _vp, _vp]),
}
# include/hbmpc_hip_debug.h: diagnostics for scratch/ scripts and white-box tests, not part of the drop-in surface
DEBUG_SYMBOLS = {
    "hb_debug_mm8_create": (_i, [_vp, _vp, _i, _i, _pp]),
    "hb_debug_mm8_apply": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "hb_debug_occupancy": (_i, [_i, _i, _vp, _vp]),
    "hb_debug_reload_env": (None, []),
}

_lib = None


def load_library():
    """dlopen the HIP library and type every declared symbol.  Raises loudly if missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64; it must be the one HIP runtime of the
    # process, so import torch (which loads it) before dlopen'ing our library, whose
    # libamdhip64 dependency then resolves to the already-loaded SONAME.
    import torch

    if torch.cuda.is_available():
        torch.cuda.init()
    if not os.path.exists(LIB_PATH):
        raise HbmpcBackendError(
            f"{LIB_PATH} not found: build it with honeybadgermpc_amd/csrc/build.sh "
            "(or __graft_entry__.build()).  There is no CPU fallback."
        )
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise HbmpcBackendError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes) in list(SYMBOLS.items()) + list(DEBUG_SYMBOLS.items()):
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    if _marshal is not None and hasattr(_marshal, "bind_dec"):
        _marshal.bind_dec(ctypes.cast(lib.hb_dec_arrived1, ctypes.c_void_p).value)      # DeviceIncrementalDecoder.add calls it without ctypes
        if hasattr(_marshal, "bind_wait"):
            _marshal.bind_wait(ctypes.cast(lib.hb_wait_arrived1, ctypes.c_void_p).value)
    return lib


# ---------------------------------------------------------------------------
# int <-> limb marshalling.  The reference's FFI wire format is little-endian
# bytes per element (hbmpc_ntl_helpers.pyx:20-29); ours is the same bytes at a
# fixed width, packed into one contiguous buffer per call.
# ---------------------------------------------------------------------------
def _load_marshal():
    """optional C helper (csrc/hb_pymarshal.c); marshalling is plumbing, so a pure-Python
    fallback for it is fine -- unlike arithmetic, which has none"""
    try:
        import importlib.util

        path = os.path.join(_HERE, "lib", "_hbmarshal.so")
        spec = importlib.util.spec_from_file_location("_hbmarshal", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    except Exception:  # noqa: BLE001
        return None


_marshal = _load_marshal()


def ints_to_limbs(values, modulus, nbytes=32):
    """list[int] -> (len, nbytes//8) uint64.  Values are reduced mod p on entry
    (pyx:31-32); negative ints raise OverflowError like int.to_bytes (pyx:20-22)."""
    if _marshal is not None:
        raw = _marshal.pack(values, modulus, nbytes)
        return np.frombuffer(bytearray(raw), dtype=np.uint64).reshape(len(values), nbytes // 8)
    buf = bytearray(len(values) * nbytes)
    off = 0
    for v in values:
        if v < 0:
            raise OverflowError("can't convert negative int to unsigned")
        if v >= modulus:
            v %= modulus
        buf[off : off + nbytes] = v.to_bytes(nbytes, "little")
        off += nbytes
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(values), nbytes // 8)


def limbs_to_ints(arr, nbytes=32):
    if _marshal is not None:
        return _marshal.unpack(np.ascontiguousarray(arr), nbytes)
    b = np.ascontiguousarray(arr).tobytes()
    return [int.from_bytes(b[i : i + nbytes], "little") for i in range(0, len(b), nbytes)]


def np_ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Context:
    """One (modulus, device) pair: owns an ``hb_ctx`` and does the torch plumbing."""

    _cache = {}
    _seen = {}          # the (modulus, device, n_limbs) triples as callers spell them -> context

    @classmethod
    def get(cls, modulus, device=None, n_limbs=None):
        """n_limbs: 1 (8-byte elements, p < 2^64) or 4 (32-byte elements); default = the narrowest that holds p."""
        # (a decoder is made per open and per round: the context it asks for is found without touching torch)
        ctx = cls._seen.get((modulus, device, n_limbs)) if device is not None else None
        if ctx is not None:
            return ctx
        import torch

        if not torch.cuda.is_available():
            raise HbmpcBackendError(
                "honeybadgermpc_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                "and there is no CPU fallback."
            )
        asked = (modulus, device, n_limbs) if device is not None else None
        if device is None:
            device = torch.cuda.current_device()
        if n_limbs is None:
            n_limbs = 1 if int(modulus) < (1 << 64) else 4
        key = (int(modulus), int(device), int(n_limbs))
        ctx = cls._cache.get(key)
        if ctx is None:
            ctx = cls(modulus, device, n_limbs)
            cls._cache[key] = ctx
        if asked is not None and isinstance(device, int) and len(cls._seen) < 64:
            cls._seen[asked] = ctx
        return ctx

    def __init__(self, modulus, device, n_limbs=4):
        import torch

        self.torch = torch
        self.lib = load_library()
        self.modulus = int(modulus)
        self.device = int(device)
        if self.modulus >= 1 << 256:
            raise ValueError("modulus must be below 2**256")
        if self.modulus % 2 == 0 or self.modulus < 3:
            raise ValueError("modulus must be an odd prime")
        # two instantiations of every kernel: 9 x 29-bit digits / 32-byte elements for p < 2^256 and
        # 3 digits / 8-byte elements for p < 2^64 (the "64/256-bit prime" of the north star)
        if n_limbs not in (1, 4) or (n_limbs == 1 and self.modulus >= 1 << 64):
            raise ValueError("n_limbs must be 4, or 1 for a modulus below 2**64")
        self.n_limbs = int(n_limbs)
        self.nbytes = 8 * self.n_limbs
        p = ints_to_limbs([self.modulus], self.modulus + 1, self.nbytes)
        h = ctypes.c_void_p()
        rc = self.lib.hb_ctx_create(ctypes.byref(h), np_ptr(p), self.n_limbs, self.device)
        if rc != HB_OK:
            raise HbmpcBackendError(f"hb_ctx_create failed: {_STATUS_NAMES.get(rc, rc)}")
        self.h = h
        self.tdev = torch.device("cuda", self.device)
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)

    # -- plumbing ----------------------------------------------------------
    def stream(self):
        """torch's CURRENT stream on this context's device, as the void* the C ABI takes (asked anew at every call: callers switch
        streams; the raw getter costs a fraction of a microsecond where building a torch.cuda.Stream object cost three)"""
        raw = self._raw_stream
        if raw is not None:
            return ctypes.c_void_p(raw(self.device))
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.tdev).cuda_stream)

    def check(self, rc, what):
        if rc == HB_OK:
            return
        msg = self.lib.hb_last_error(self.h)
        raise HbmpcBackendError(f"{what}: {_STATUS_NAMES.get(rc, rc)}: {msg.decode() if msg else ''}")

    def empty(self, count):
        """device buffer of `count` elements, as an int64 tensor of shape (count, n_limbs)"""
        return self.torch.empty((max(int(count), 0), self.n_limbs), dtype=self.torch.int64, device=self.tdev)

    def to_device(self, limbs):
        t = self.torch.from_numpy(np.ascontiguousarray(limbs).view(np.int64))
        return t.to(self.tdev)

    def upload_ints(self, values):
        return self.to_device(ints_to_limbs(values, self.modulus, self.nbytes))

    def download_ints(self, tensor):
        return limbs_to_ints(tensor.cpu().numpy().view(np.uint64), self.nbytes)

    def host_elems(self, values):
        """small host-side element array (points, omega) for by-value C arguments"""
        return ints_to_limbs(values, self.modulus, self.nbytes)

    @staticmethod
    def ptr(tensor):
        return ctypes.c_void_p(tensor.data_ptr())

    def elems(self, tensor, count=None, what="tensor"):
        """Validate a caller-supplied element buffer before its data_ptr() goes to C: int64, trailing dimension
        n_limbs, on this context's device, `count` elements (when given); returns it contiguous."""
        t = self.torch
        if not isinstance(tensor, t.Tensor):
            raise TypeError(f"{what}: expected a torch tensor, got {type(tensor).__name__}")
        if tensor.dtype != t.int64:
            raise TypeError(f"{what}: dtype must be int64 (4 x u64 limbs viewed as int64), got {tensor.dtype}")
        if tensor.dim() < 1 or tensor.shape[-1] != self.n_limbs:
            raise ValueError(f"{what}: last dimension must be {self.n_limbs} limbs, got shape {tuple(tensor.shape)}")
        if not tensor.is_cuda or tensor.device.index != self.device:
            raise ValueError(f"{what}: must live on cuda:{self.device}, is on {tensor.device}")
        if count is not None and tensor.numel() != int(count) * self.n_limbs:
            raise ValueError(f"{what}: expected {int(count)} elements, got {tensor.numel() // self.n_limbs}")
        return tensor if tensor.is_contiguous() else tensor.contiguous()

    def reduce_(self, tensor):
        """in place: every element of a (count, limbs) device tensor -> its canonical residue (hb_reduce; the reference reduces
        whatever enters its boundary, pyx:31-32).  For buffers filled from outside: wire payloads, files."""
        tensor = self.elems(tensor, what="tensor")
        count = tensor.numel() // self.n_limbs
        self.check(self.lib.hb_reduce(self.h, self.ptr(tensor), self.ptr(tensor), count, None, self.stream()), "hb_reduce")
        return tensor

    def cache_clear(self):
        """drop every cached table of this context (hb_ctx_cache_clear)"""
        self.check(self.lib.hb_ctx_cache_clear(self.h), "hb_ctx_cache_clear")

    def cache_entries(self):
        return int(self.lib.hb_ctx_cache_entries(self.h))
