"""The C-ABI boundary defends itself:
  * the pipelined entry points (kzg_commit_device_async / kzg_open_device_async) keep raw host pointers until
    kzg_commit_flush: the Python context holds the arrays (and the key) for exactly that long;
  * kzg_ctx_set_stream refuses what cannot be a stream handle (HIP cannot validate one: the caller guarantees a live
    stream, the library asks nothing of the runtime);
  * a failed opening leaves no stale evaluation pointer in its pipeline slot."""
import gc
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipelined_outputs_must_be_real_buffers():
    from kzg_snark_amd import _native
    ok = np.zeros(12, dtype=np.uint64)
    _native._check_out(ok, np.uint64, 12, "out_xy")
    _native._check_out(None, np.uint64, 4, "eval_out")                 # the evaluation is optional
    for bad in ([0] * 12, np.zeros(12, dtype=np.int32), np.zeros(11, dtype=np.uint64),
                np.zeros(24, dtype=np.uint64)[::2], None):
        with pytest.raises(TypeError):
            _native._check_out(bad, np.uint64, 12, "out_xy")
    ro = np.zeros(12, dtype=np.uint64)
    ro.flags.writeable = False
    with pytest.raises(TypeError):
        _native._check_out(ro, np.uint64, 12, "out_xy")


@pytest.mark.gpu
def test_context_holds_pipelined_outputs_until_the_flush(native):
    """Drop every caller-side reference between the enqueue and the flush: the results still land in live memory
    (the context's list), and equal the synchronous call's."""
    import torch
    from oracle import py_oracle as O
    cv = O.BLS12_381
    ctx = native.get_context("bls12_381")
    n = 1 << 14
    tau = 0xabcdef12345
    srs = ctx.srs_generate(native.int_to_words(tau), n)
    rs = np.random.RandomState(3)
    raw = rs.randint(0, 1 << 62, size=(3, n, 4)).astype(np.uint64)
    raw[:, :, 3] >>= np.uint64(3)
    d = torch.from_numpy(raw.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    want_xy, want_inf = ctx.commit_device(srs, d.data_ptr(), [n, n, n], n)
    L = ctx.fp_limbs
    assert ctx._inflight == []
    ctx.commit_device_async(srs, d.data_ptr(), [n, n, n], n, np.zeros((3, 2 * L), dtype=np.uint64),
                            np.zeros(3, dtype=np.uint8))                # temporaries: nobody else holds them
    z, xi = native.int_to_words(12345), native.int_to_words(678)
    ctx.open_device_async(srs, d.data_ptr(), [n, n, n], n, z, xi, np.zeros(2 * L, dtype=np.uint64),
                          np.zeros(1, dtype=np.uint8), np.zeros(4, dtype=np.uint64))
    assert len(ctx._inflight) == 2
    held = list(ctx._inflight)
    gc.collect()
    junk = [np.full(2 * L * 3, 0xdeadbeef, dtype=np.uint64) for _ in range(64)]     # would reuse freed blocks
    ctx.commit_flush()
    assert ctx._inflight == []
    assert np.array_equal(held[0][1], want_xy) and np.array_equal(held[0][2], want_inf)
    oxy, oinf, ev = ctx.open(srs, d.data_ptr(), [n, n, n], n, z, xi, device=True)
    assert np.array_equal(held[1][1], oxy) and np.array_equal(held[1][3], ev)
    assert all(int(j[0]) == 0xdeadbeef for j in junk)
    with pytest.raises(TypeError):
        ctx.commit_device_async(srs, d.data_ptr(), [n], n, [0] * (2 * L), np.zeros(1, dtype=np.uint8))
    assert ctx._inflight == []
    srs.close()


@pytest.mark.gpu
def test_failed_open_leaves_no_stale_evaluation_pointer(native):
    """an opening that fails after its P(z) copy was queued (here: a polynomial longer than the key)
    must not leave its eval_out in the slot for the next commit to write through."""
    import torch
    ctx = native.get_context("bls12_381")
    n = 1 << 10
    srs = ctx.srs_generate(native.int_to_words(77), n)
    d = torch.zeros((1, 2 * n, 4), dtype=torch.int64, device="cuda:0")
    d[0, :, 0] = 1
    torch.cuda.synchronize()
    L = ctx.fp_limbs
    ev = np.full(4, 0x5a5a5a5a, dtype=np.uint64)
    with pytest.raises(native.NativeError):
        ctx.open_device_async(srs, d.data_ptr(), [2 * n], 2 * n, native.int_to_words(3), native.int_to_words(5),
                              np.zeros(2 * L, dtype=np.uint64), np.zeros(1, dtype=np.uint8), ev)
    ev[:] = 0x5a5a5a5a          # (the call zeroes eval_out on entry, while the caller's buffer is certainly alive)
    ctx.commit_flush()
    for _ in range(6):                                                 # walk over every pipeline slot
        ctx.commit_device(srs, d.data_ptr(), [n], 2 * n)
    assert np.all(ev == 0x5a5a5a5a)
    srs.close()


@pytest.mark.gpu
def test_set_stream_refuses_what_cannot_be_a_stream():
    """gpurun_out/r02_crash.log: a non-handle passed to kzg_ctx_set_stream used to reach hipEventRecord.  HIP cannot
    validate a handle (hipStreamQuery dereferences it: a readable buffer that is no stream crashed the child of this
    test's first version, profiles/r03_stream_query_crash.log; round 4 dropped the query altogether), so the library
    refuses what cannot be a runtime object -- small integers other than the two documented aliases, misaligned
    values -- and takes real streams and the aliases; liveness of anything else is the caller's guarantee.  Run in a child process: a regression here is a host crash, which must not take the session down."""
    code = r'''
import sys
sys.path.insert(0, %r)
import torch
from kzg_snark_amd import _native
ctx = _native.Context("bls12_381")
s = torch.cuda.Stream()
refused = 0
for bogus in (3, 7, 0xdead, s.cuda_stream + 1, s.cuda_stream + 4):
    try:
        ctx.set_stream(bogus)
        print("ADOPTED", hex(bogus))
    except _native.NativeError as e:
        refused += 1
print("REFUSED", refused)
ctx.set_stream(s.cuda_stream)                                # a real stream is taken
ctx.bind_torch_stream(torch.cuda.default_stream())           # so is the null stream's explicit name (hipStreamLegacy)
ctx.set_stream(2)                                            # and hipStreamPerThread
ctx.set_stream(0)                                            # back to a private stream
x = torch.zeros((16, 4), dtype=torch.int64, device="cuda:0"); x[:, 0] = 1
torch.cuda.synchronize()
ctx.ntt_device(x.data_ptr(), 4, _native.int_to_words(1), False, 1)
ctx.synchronize()
print("SUM", int(x[0, 0]))
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-2000:])
    lines = out.stdout.split()
    assert "ADOPTED" not in lines and lines[lines.index("REFUSED") + 1] == "5", out.stdout
    assert lines[-2:] == ["SUM", "16"], out.stdout               # w = 1: X[0] = sum of sixteen ones


def test_default_device_follows_the_environment(monkeypatch):
    """One process per GPU: the facade's device is KZG_MI355X_DEVICE, else torch's current device once torch.cuda is up
    (not on this CPU box), else 0."""
    from kzg_snark_amd import _native
    monkeypatch.delenv("KZG_MI355X_DEVICE", raising=False)
    import torch
    if not (torch.cuda.is_available() and torch.cuda.is_initialized()):
        assert _native.default_device() == 0
    monkeypatch.setenv("KZG_MI355X_DEVICE", "5")
    assert _native.default_device() == 5


@pytest.mark.gpu
def test_tuning_keys_and_values_are_checked():
    """kzg_ctx_set_tuning refuses unknown keys and values outside what the kernels are built for; 0 restores the
    library's own choice.  (Results never depend on a tuning value: tests/test_kzg_gpu.py, tests/test_ntt_gpu.py.)"""
    from kzg_snark_amd import _native
    ctx = _native.get_context("bls12_381")
    for key, bad in (("ntt_tile_log", 7), ("ntt_tile_log", 13), ("open_tile_threads", 64), ("open_tile_threads", 512),
                     ("open_direct_tiles", -1), ("no_such_key", 1)):
        with pytest.raises(_native.NativeError):
            ctx.set_tuning(key, bad)
    for key, ok in (("ntt_tile_log", 10), ("open_tile_threads", 128), ("open_direct_tiles", 5)):
        ctx.set_tuning(key, ok)
        ctx.set_tuning(key, 0)


@pytest.mark.gpu
def test_partial_transform_passes_check_their_ranges():
    """The local halves of the distributed transform refuse sizes the tiled kernels do not take (log_n <= 12), counts
    that are not powers of two, and line ranges beyond the matrix -- before anything is launched."""
    import torch
    from kzg_snark_amd import _native
    ctx = _native.get_context("bls12_381")
    r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    buf = torch.zeros((1 << 14, 4), dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    w14, w12 = _native.int_to_words(pow(7, (r - 1) >> 14, r)), _native.int_to_words(pow(7, (r - 1) >> 12, r))
    p = buf.data_ptr()
    for call in (lambda: ctx.ntt_columns_device(p, 12, w12, False, 64, 0),              # single-tile size
                 lambda: ctx.ntt_columns_device(p, 14, w14, False, 96, 0),              # not a power of two
                 lambda: ctx.ntt_columns_device(p, 14, w14, False, 64, 96),             # columns 96 .. 159 of 128
                 lambda: ctx.ntt_rows_device(p, 14, w14, False, 256),                   # more rows than the matrix has
                 lambda: ctx.ntt_rows_twist_device(p, 14, w14, False, 64, 96),          # rows 96 .. 159 of 128
                 lambda: ctx.ntt_columns_plain_device(p, 14, w14, False, 256)):         # more columns than it has
        with pytest.raises(_native.NativeError):
            call()
    ctx.ntt_rows_twist_device(p, 14, w14, False, 64, 64)                                # rows 64 .. 127: fine
    ctx.synchronize()
