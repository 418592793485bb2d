"""ctypes binding of libkzg_mi355x.so (include/kzg_mi355x.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device
is visible, the first call raises NativeUnavailable -- loudly, by design."""
import ctypes
import os
from operator import methodcaller as _methodcaller

# The commit pipeline runs three internal streams beside the caller's; HIP multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise.
# Must be in the environment before the HIP runtime initialises (import this module -- or set
# the variable -- before the first torch.cuda / HIP call).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# KZG_MI355X_LIB: load another build of the library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("KZG_MI355X_LIB") or os.path.join(_HERE, "lib", "libkzg_mi355x.so")

KZG_CURVE_BN254 = 0
KZG_CURVE_BLS12_381 = 1
CURVE_IDS = {"bn254": KZG_CURVE_BN254, "bls12_381": KZG_CURVE_BLS12_381}

HIP_STREAM_LEGACY = 1          # hipStreamLegacy: the explicit handle of HIP's null stream (hip_runtime_api.h)
KZG_ERR_DEGREE = -4
KZG_ERR_NODEV = -3

# every symbol include/kzg_mi355x.h declares (tests/test_abi.py checks the .so exports them all)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p
SIGNATURES = {
    "kzg_abi_version": (ctypes.c_int, []),
    "kzg_fp_limbs": (ctypes.c_int, [ctypes.c_int]),
    "kzg_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    "kzg_ctx_destroy": (None, [_vp]),
    "kzg_last_error": (ctypes.c_char_p, [_vp]),
    "kzg_ctx_set_stream": (ctypes.c_int, [_vp, _vp]),
    "kzg_fft_ff_any": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_int]),
    "kzg_fft_ff_any_device": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.c_int]),
    "kzg_ctx_synchronize": (ctypes.c_int, [_vp]),
    "kzg_ctx_set_tuning": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int64]),
    "kzg_ntt": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp, ctypes.c_int]),
    "kzg_ntt_device": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp, ctypes.c_int, ctypes.c_uint32]),
    "kzg_ntt_columns_device": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp, ctypes.c_int, ctypes.c_uint64,
                                              ctypes.c_uint64]),
    "kzg_ntt_rows_device": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp, ctypes.c_int, ctypes.c_uint64]),
    "kzg_ntt_rows_twist_device": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64]),
    "kzg_ntt_columns_plain_device": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp, ctypes.c_int, ctypes.c_uint64]),
    "kzg_ntt_rows_exchange_device": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_int, ctypes.c_uint64,
                                                    ctypes.c_uint32, ctypes.c_int]),
    "kzg_srs_generate_strided": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                                ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "kzg_srs_load_g1": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "kzg_srs_free": (None, [_vp]),
    "kzg_srs_size": (ctypes.c_size_t, [_vp]),
    "kzg_srs_generate": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "kzg_srs_generate_range": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "kzg_srs_export": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp]),
    "kzg_commit": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp]),
    "kzg_commit_device": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp]),
    "kzg_commit_device_async": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp]),
    "kzg_commit_flush": (ctypes.c_int, [_vp]),
    "kzg_open": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp, _vp, _vp, _vp]),
    "kzg_open_shard_begin": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp, _vp]),
    "kzg_open_shard_finish": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "kzg_fr_vec_op": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_size_t, _vp, _vp, _vp]),
    "kzg_fr_vec_lincomb": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp, _vp, _vp]),
    "kzg_fr_vec_mul_powers": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp, _vp, _vp]),
    "kzg_fr_vec_inverse": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp]),
    "kzg_fr_vec_prefix_product": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp]),
    "kzg_fr_poly_eval": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp, _vp]),
    "kzg_prof_enable": (ctypes.c_int, [_vp, ctypes.c_int]),
    "kzg_prof_reset": (ctypes.c_int, [_vp]),
    "kzg_prof_read": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_uint64)]),
    "kzg_open_device_async": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp, _vp, _vp, _vp]),
    "kzg_g1_sum": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_size_t, _vp, _vp]),
    "kzg_open_device": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp, _vp, _vp, _vp]),
}


class NativeUnavailable(RuntimeError):
    pass


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libkzg_mi355x error {code}: {msg}")
        self.code = code


_lib = None
MISSING = []


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  If this
    library were loaded first it would bind the system runtime and a later `import torch` would
    start a SECOND HIP runtime in the process (torch then reports "No HIP GPUs are available", and
    stream handles could not be shared).  Loading torch's copy first -- by path, without importing
    torch -- makes both sides resolve to one runtime whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load the shared library (no GPU needed for loading; needed for contexts)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeUnavailable(
                f"{LIB_PATH} not built: run `python -m kzg_snark_amd.build` (no CPU fallback exists)")
        _preload_torch_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is None:          # tests/test_abi.py requires MISSING to stay empty
                MISSING.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check_out(a, dtype, size, what):
    """An output of a pipelined call is written later through a raw pointer: it must be a writable, contiguous numpy
    array of the right size (a temporary or a list would be gone, or never seen, by then)."""
    if a is None and what == "eval_out":
        return
    if not (isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous and a.flags.writeable
            and a.size >= size):
        raise TypeError(f"{what}: need a writable C-contiguous numpy {np.dtype(dtype).name} array of >= {size} elements")


def _as_vp(a):
    """numpy array / int device pointer / ctypes array -> c_void_p."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(ctypes.c_void_p)
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    return ctypes.cast(a, ctypes.c_void_p)


class Context:
    """One engine context: one curve, one GPU, one stream (kzg_ctx)."""

    def __init__(self, curve_type="bls12_381", device=0):
        if curve_type not in CURVE_IDS:
            raise ValueError(f"Unsupported curve type: {curve_type}")    # kzg.py:37
        self.curve_type = curve_type
        self.device = int(device)
        self.curve_id = CURVE_IDS[curve_type]
        self.fp_limbs = lib().kzg_fp_limbs(self.curve_id)
        h = ctypes.c_void_p()
        rc = lib().kzg_ctx_create(self.curve_id, int(device), ctypes.byref(h))
        if rc == KZG_ERR_NODEV:
            raise NativeUnavailable(
                "kzg_ctx_create: no gfx950 (MI355X) device visible; this engine has no CPU fallback")
        if rc != 0:
            raise NativeError(rc, "kzg_ctx_create failed")
        self._h = h
        # host arrays (and keys) the library still holds raw pointers to: the outputs of the pipelined entry points
        # are written when their slot is retired -- by a later commit or by commit_flush() -- so the context keeps
        # them alive until the flush, whatever the caller does with its own references
        self._inflight = []

    def close(self):
        if getattr(self, "_h", None):
            lib().kzg_ctx_destroy(self._h)       # waits for the device; pending results are dropped, not written
            self._h = None
            self._inflight = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = lib().kzg_last_error(self._h)
            raise NativeError(rc, msg.decode() if msg else "")

    def set_stream(self, stream_ptr):
        self._check(lib().kzg_ctx_set_stream(self._h, ctypes.c_void_p(stream_ptr or 0)))

    def bind_torch_stream(self, stream=None):
        """Run this context's work on a torch stream (default: torch's CURRENT stream on the
        context's device) so that tensors produced or consumed by torch ops are ordered with the
        engine's kernels without explicit synchronisation.  A fresh context owns a private
        non-blocking stream: a torch producer on another stream then needs torch.cuda.synchronize()
        (or an event) before the engine reads its output -- INTEGRATION.md, "stream ordering".
        Returns the torch stream that was bound."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        # torch's default stream is HIP's null stream (handle 0), and a NULL handle means "back to the
        # context's own stream" in kzg_ctx_set_stream: name the null stream explicitly instead
        self.set_stream(stream.cuda_stream or HIP_STREAM_LEGACY)
        self._torch_stream = stream            # keep the handle alive as long as it is bound
        return stream

    def synchronize(self):
        self._check(lib().kzg_ctx_synchronize(self._h))

    def set_tuning(self, key, value):
        """Fix a choice the library otherwise makes from what is resident ("ntt_tile_log", "open_tile_threads",
        "open_direct_tiles"; 0 = the library's choice).  Results never depend on it."""
        self._check(lib().kzg_ctx_set_tuning(self._h, key.encode(), int(value)))

    # ---- measurement hooks
    def prof_enable(self, on=True):
        self._check(lib().kzg_prof_enable(self._h, int(bool(on))))

    def prof_reset(self):
        self._check(lib().kzg_prof_reset(self._h))

    def prof_read(self, name):
        """(total milliseconds, launches) of one span since the last reset."""
        ms, cnt = ctypes.c_double(0), ctypes.c_uint64(0)
        self._check(lib().kzg_prof_read(self._h, name.encode(), ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    # ---- NTT
    def ntt(self, data, log_n, w_words, inverse):
        """data: C-contiguous uint64[n,4] numpy array, transformed in place."""
        assert data.dtype == np.uint64 and data.flags.c_contiguous and data.size == 4 << log_n
        self._check(lib().kzg_ntt(self._h, _as_vp(data), log_n, _as_vp(w_words), int(bool(inverse))))

    def fft_ff_any(self, data, w_words, inverse):
        """fft_ff / ifft_ff of a list of ANY length >= 1 (reference recursion semantics), in place."""
        assert data.dtype == np.uint64 and data.flags.c_contiguous and data.ndim == 2 and data.shape[1] == 4
        self._check(lib().kzg_fft_ff_any(self._h, _as_vp(data), data.shape[0], _as_vp(w_words), int(bool(inverse))))

    def ntt_device(self, d_ptr, log_n, w_words, inverse, batch=1):
        self._check(lib().kzg_ntt_device(self._h, _as_vp(d_ptr), log_n, _as_vp(w_words),
                                         int(bool(inverse)), batch))

    def ntt_columns_device(self, d_ptr, log_n, w_words, inverse, n_cols, col_base):
        self._check(lib().kzg_ntt_columns_device(self._h, _as_vp(d_ptr), log_n, _as_vp(w_words),
                                                 int(bool(inverse)), n_cols, col_base))

    def ntt_rows_device(self, d_ptr, log_n, w_words, inverse, n_rows):
        self._check(lib().kzg_ntt_rows_device(self._h, _as_vp(d_ptr), log_n, _as_vp(w_words),
                                              int(bool(inverse)), n_rows))

    def ntt_rows_twist_device(self, d_ptr, log_n, w_words, inverse, n_rows, row_base):
        self._check(lib().kzg_ntt_rows_twist_device(self._h, _as_vp(d_ptr), log_n, _as_vp(w_words),
                                                    int(bool(inverse)), n_rows, row_base))

    def ntt_columns_plain_device(self, d_ptr, log_n, w_words, inverse, n_cols):
        self._check(lib().kzg_ntt_columns_plain_device(self._h, _as_vp(d_ptr), log_n, _as_vp(w_words),
                                                       int(bool(inverse)), n_cols))

    def ntt_rows_exchange_device(self, d_src, d_dst, log_n, w_words, inverse, n_rows, world, blocked_out):
        self._check(lib().kzg_ntt_rows_exchange_device(self._h, _as_vp(d_src), _as_vp(d_dst), log_n, _as_vp(w_words),
                                                       int(bool(inverse)), n_rows, world, int(bool(blocked_out))))

    # ---- SRS
    def srs_load_g1(self, xy, inf=None):
        """xy: uint64[n, 2*fp_limbs] affine canonical; inf: uint8[n] flags or None."""
        n = xy.shape[0]
        assert xy.dtype == np.uint64 and xy.flags.c_contiguous and xy.shape[1] == 2 * self.fp_limbs
        if inf is not None:
            assert inf.dtype == np.uint8 and inf.size == n and inf.flags.c_contiguous
        h = ctypes.c_void_p()
        self._check(lib().kzg_srs_load_g1(self._h, _as_vp(xy), _as_vp(inf), n, ctypes.byref(h)))
        return Srs(self, h, n)

    def srs_generate(self, tau_words, n, start=0):
        h = ctypes.c_void_p()
        self._check(lib().kzg_srs_generate_range(self._h, _as_vp(tau_words), start, n, ctypes.byref(h)))
        return Srs(self, h, n)

    def srs_generate_strided(self, tau_words, start, n, run_len, inner_stride, outer_stride):
        """point i = tau^(start + (i // run_len) * outer_stride + (i % run_len) * inner_stride) * G1"""
        h = ctypes.c_void_p()
        self._check(lib().kzg_srs_generate_strided(self._h, _as_vp(tau_words), start, n, run_len, inner_stride,
                                                   outer_stride, ctypes.byref(h)))
        return Srs(self, h, n)

    # ---- commit / open on host buffers
    def commit(self, srs, scalars, lens, stride):
        """scalars: uint64[n_polys, stride, 4]; lens: per-polynomial coefficient counts."""
        n_polys = len(lens)
        lens_a = np.asarray(lens, dtype=np.uint64)
        out_xy = np.zeros((n_polys, 2 * self.fp_limbs), dtype=np.uint64)
        out_inf = np.zeros(n_polys, dtype=np.uint8)
        self._check(lib().kzg_commit(self._h, srs._h, _as_vp(scalars), _as_vp(lens_a), n_polys, stride,
                                     _as_vp(out_xy), _as_vp(out_inf)))
        return out_xy, out_inf

    def commit_device(self, srs, d_scalars, lens, stride):
        n_polys = len(lens)
        lens_a = np.asarray(lens, dtype=np.uint64)
        out_xy = np.zeros((n_polys, 2 * self.fp_limbs), dtype=np.uint64)
        out_inf = np.zeros(n_polys, dtype=np.uint8)
        self._check(lib().kzg_commit_device(self._h, srs._h, _as_vp(d_scalars), _as_vp(lens_a), n_polys,
                                            stride, _as_vp(out_xy), _as_vp(out_inf)))
        return out_xy, out_inf

    def commit_device_async(self, srs, d_scalars, lens, stride, out_xy, out_inf):
        """Pipelined commit: results land in the caller's out_xy / out_inf (numpy, kept alive by the
        caller) by the time commit_flush() returns."""
        lens_a = np.asarray(lens, dtype=np.uint64)
        _check_out(out_xy, np.uint64, len(lens) * 2 * self.fp_limbs, "out_xy")
        _check_out(out_inf, np.uint8, len(lens), "out_inf")
        # the library adopts the host pointers only once the work is queued (msm.hip): on an error nothing points
        # at these arrays, so they are recorded after the call has returned KZG_OK
        self._check(lib().kzg_commit_device_async(self._h, srs._h, _as_vp(d_scalars), _as_vp(lens_a), len(lens),
                                                  stride, _as_vp(out_xy), _as_vp(out_inf)))
        self._inflight.append((srs, out_xy, out_inf))

    def commit_flush(self):
        """Drain the pipeline: every pending result is on the host when this returns (or raises)."""
        try:
            self._check(lib().kzg_commit_flush(self._h))
        finally:
            # the library retires every slot in kzg_commit_flush, also on error: nothing points at these any more
            self._inflight.clear()

    def open(self, srs, polys, lens, stride, z_words, xi_words, device=False):
        k = len(lens)
        lens_a = np.asarray(lens, dtype=np.uint64)
        out_xy = np.zeros(2 * self.fp_limbs, dtype=np.uint64)
        out_inf = np.zeros(1, dtype=np.uint8)
        ev = np.zeros(4, dtype=np.uint64)
        fn = lib().kzg_open_device if device else lib().kzg_open
        self._check(fn(self._h, srs._h, _as_vp(polys), _as_vp(lens_a), k, stride, _as_vp(z_words),
                       _as_vp(xi_words), _as_vp(out_xy), _as_vp(out_inf), _as_vp(ev)))
        return out_xy, out_inf, ev


    def open_device_async(self, srs, d_polys, lens, stride, z_words, xi_words, out_xy, out_inf, eval_out):
        """Pipelined open: out_xy (uint64[2*fp_limbs]), out_inf (uint8[1]) and eval_out (uint64[4]) -- numpy arrays
        the caller keeps alive -- are filled by the time commit_flush() returns."""
        lens_a = np.asarray(lens, dtype=np.uint64)
        _check_out(out_xy, np.uint64, 2 * self.fp_limbs, "out_xy")
        _check_out(out_inf, np.uint8, 1, "out_inf")
        _check_out(eval_out, np.uint64, 4, "eval_out")
        self._check(lib().kzg_open_device_async(self._h, srs._h, _as_vp(d_polys), _as_vp(lens_a), len(lens), stride,
                                                _as_vp(z_words), _as_vp(xi_words), _as_vp(out_xy), _as_vp(out_inf),
                                                _as_vp(eval_out)))
        self._inflight.append((srs, out_xy, out_inf, eval_out))

    # ---- device vector / polynomial primitives (device pointers, canonical elements)
    def vec_op(self, op, n, d_a, d_b, d_out):
        self._check(lib().kzg_fr_vec_op(self._h, {"add": 0, "sub": 1, "mul": 2}[op], n, _as_vp(d_a), _as_vp(d_b),
                                        _as_vp(d_out)))

    def vec_lincomb(self, n, d_ptrs, lens, scalars, d_out):
        k = len(d_ptrs)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[int(p) for p in d_ptrs])
        lens_a = np.asarray(lens, dtype=np.uint64)
        sc = np.ascontiguousarray(np.concatenate([int_to_words(int(s)) for s in scalars]) if k else np.zeros(4, np.uint64))
        self._check(lib().kzg_fr_vec_lincomb(self._h, n, k, ctypes.cast(ptrs, ctypes.c_void_p), _as_vp(lens_a),
                                             _as_vp(sc), _as_vp(d_out)))

    def vec_mul_powers(self, n, d_a, s, c0, d_out):
        self._check(lib().kzg_fr_vec_mul_powers(self._h, n, _as_vp(d_a), _as_vp(int_to_words(int(s))),
                                                _as_vp(int_to_words(int(c0))), _as_vp(d_out)))

    def vec_inverse(self, n, d_a, d_out):
        self._check(lib().kzg_fr_vec_inverse(self._h, n, _as_vp(d_a), _as_vp(d_out)))

    def vec_prefix_product(self, n, d_a, d_out):
        self._check(lib().kzg_fr_vec_prefix_product(self._h, n, _as_vp(d_a), _as_vp(d_out)))

    def poly_eval(self, n, d_a, z):
        out = np.zeros(4, dtype=np.uint64)
        self._check(lib().kzg_fr_poly_eval(self._h, n, _as_vp(d_a), _as_vp(int_to_words(int(z))), _as_vp(out)))
        return int.from_bytes(out.tobytes(), "little")

    # ---- sharded open (device pointers)
    def open_shard_begin(self, d_polys, lens, stride, z_words, xi_words):
        lens_a = np.asarray(lens, dtype=np.uint64)
        h = np.zeros(4, dtype=np.uint64)
        self._check(lib().kzg_open_shard_begin(self._h, _as_vp(d_polys), _as_vp(lens_a), len(lens), stride,
                                               _as_vp(z_words), _as_vp(xi_words), _as_vp(h)))
        return h

    def open_shard_finish(self, srs, z_words, carry_words, first_rank):
        out_xy = np.zeros(2 * self.fp_limbs, dtype=np.uint64)
        out_inf = np.zeros(1, dtype=np.uint8)
        ev = np.zeros(4, dtype=np.uint64)
        self._check(lib().kzg_open_shard_finish(self._h, srs._h, _as_vp(z_words), _as_vp(carry_words),
                                                int(bool(first_rank)), _as_vp(out_xy), _as_vp(out_inf), _as_vp(ev)))
        return out_xy, out_inf, ev


class Srs:
    """Device-resident commitment key (kzg_srs): the reference's `ck` list."""

    def __init__(self, ctx, h, n):
        self.ctx = ctx
        self._h = h
        self.n = n

    def export(self, start=0, count=None):
        count = self.n - start if count is None else count
        xy = np.zeros((count, 2 * self.ctx.fp_limbs), dtype=np.uint64)
        inf = np.zeros(count, dtype=np.uint8)
        self.ctx._check(lib().kzg_srs_export(self.ctx._h, self._h, start, count, _as_vp(xy), _as_vp(inf)))
        return xy, inf

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            lib().kzg_srs_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def g1_sum(curve_type, points):
    """Sum of G1 points given as facade tuples (x, y, 1) / (1, 1, 0), on the host in the library's C++ (kzg_g1_sum):
    microseconds per point where the pure-Python group law takes tens -- the add-up of the ranks' partial results."""
    cid = CURVE_IDS[curve_type]
    L = lib().kzg_fp_limbs(cid)
    n = len(points)
    inf = np.array([1 if int(p[2]) == 0 else 0 for p in points], dtype=np.uint8)
    coords = []
    for p, f in zip(points, inf):
        coords += [0, 0] if f else [int(p[0]), int(p[1])]
    xy = ints_to_limbs(coords, L).reshape(n, 2 * L) if n else np.zeros((0, 2 * L), dtype=np.uint64)
    out_xy, out_inf = np.zeros(2 * L, dtype=np.uint64), np.zeros(1, dtype=np.uint8)
    rc = lib().kzg_g1_sum(cid, _as_vp(np.ascontiguousarray(xy)), _as_vp(inf), n, _as_vp(out_xy), _as_vp(out_inf))
    if rc != 0:
        raise NativeError(rc, "kzg_g1_sum: a coordinate is not reduced or a point is not on the curve")
    if out_inf[0]:
        return (1, 1, 0)
    v = limbs_to_ints(out_xy.reshape(2, L))
    return (v[0], v[1], 1)


_contexts = {}


def default_device():
    """The GPU the facade uses when none is named: KZG_MI355X_DEVICE, else torch's current device when this process
    has already initialised torch.cuda (one process per GPU: the launcher's torch.cuda.set_device(local_rank)), else 0."""
    import sys
    env = os.environ.get("KZG_MI355X_DEVICE")
    if env:
        return int(env)
    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                return int(torch.cuda.current_device())
        except Exception:    # noqa: BLE001 -- a torch without a usable GPU runtime: the engine will say so itself
            pass
    return 0


def get_context(curve_type, device=None):
    device = default_device() if device is None else int(device)
    key = (curve_type, device)
    if key not in _contexts:
        _contexts[key] = Context(curve_type, device)
    return _contexts[key]


# ---- integer <-> limb marshalling -------------------------------------------------

def _load_pyconv():
    """csrc/pyconv.c built by kzg_snark_amd.build (CPython API; marshalling only).  Absent: the Python forms below."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_kzg_pyconv.so")
    if not os.path.exists(path):
        return None
    try:
        import importlib.machinery
        import importlib.util
        loader = importlib.machinery.ExtensionFileLoader("_kzg_pyconv", path)
        spec = importlib.util.spec_from_loader("_kzg_pyconv", loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        return mod
    except Exception:   # noqa: BLE001 -- built for another interpreter, say: the Python forms still work
        return None


_pyconv = _load_pyconv()


def ints_to_limbs(values, limbs=4):
    """list of non-negative ints (< 2^(64*limbs)) -> uint64[n, limbs] (little-endian)."""
    nb = 8 * limbs
    if _pyconv is not None:
        if not isinstance(values, (list, tuple)):
            values = list(values)
        buf = _pyconv.ints_to_bytes(values, nb)          # a bytearray: wrapped, not copied
        return np.frombuffer(buf, dtype="<u8").reshape(len(values), limbs)
    try:                                   # plain ints (what every facade path passes): 1.8x the generator form
        buf = b"".join(map(_methodcaller("to_bytes", nb, "little"), values))
    except AttributeError:                 # field elements and other int()-able objects
        buf = b"".join(int(v).to_bytes(nb, "little") for v in values)
    return np.frombuffer(buf, dtype="<u8").reshape(len(values), limbs).copy()


def limbs_to_ints(arr):
    a = np.ascontiguousarray(arr, dtype="<u8")
    nb = 8 * a.shape[-1]
    if _pyconv is not None:
        return _pyconv.bytes_to_ints(a.reshape(-1).view(np.uint8), nb)
    raw = a.tobytes()
    return [int.from_bytes(raw[i:i + nb], "little") for i in range(0, len(raw), nb)]


def int_to_words(v, limbs=4):
    return np.frombuffer(int(v).to_bytes(8 * limbs, "little"), dtype="<u8").copy()
