"""CPU stand-ins for the engine, built on the oracle (test scaffolding; the product never imports this):
  * OracleNttOps        -- the local halves of sharding.DistributedNTT on Python ints;
  * OracleShardBackend  -- the algebra / MSM / opening backend of plonk_sharded.ShardedProver, so that the
                           vector-sharded prover's choreography runs under gloo without a GPU.
Vectors are int64 [m, 4] CPU tensors of canonical little-endian Fr elements, as on the device."""
import numpy as np
import torch

from oracle import py_oracle as O


def ints_of(t):
    a = np.ascontiguousarray(t.contiguous().numpy().view(np.uint64))
    raw = a.tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def tensor_of(vals):
    buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
    return torch.from_numpy(np.frombuffer(buf, dtype="<i8").reshape(len(vals), 4).copy())


def put(t, vals):
    t.copy_(tensor_of(vals).view(t.shape))


class OracleNttOps:
    """Local halves of the distributed NTT on Python ints (the oracle's fft_ff)."""

    def __init__(self, log_n, w, r, inverse):
        self.k1 = (log_n + 1) // 2
        self.k2 = log_n - self.k1
        self.r = r
        self.n = 1 << log_n
        self.w = pow(w, -1, r) if inverse else w
        self.scale = pow(self.n, -1, r) if inverse else 1

    _ints = staticmethod(ints_of)
    _put = staticmethod(put)

    def columns(self, M, col_base):
        N1, W = M.shape[0], M.shape[1]
        vals = ints_of(M)
        root = pow(self.w, 1 << self.k2, self.r)
        out = [0] * (N1 * W)
        for c in range(W):
            col = O.fft_ff([vals[t * W + c] for t in range(N1)], root, self.r)
            for t in range(N1):
                out[t * W + c] = col[t] * pow(self.w, t * (col_base + c), self.r) % self.r
        put(M, out)

    def rows_exchange(self, recv, out, world, blocked):
        """recv: [G][R1][W] blocks as the columns -> rows all-to-all delivers them; out: the same
        blocked shape over the output index (blocked) or plain [R1][N2] rows."""
        G, R1, W = recv.shape[0], recv.shape[1], recv.shape[2]
        assert G == world
        N2 = G * W
        vals = ints_of(recv)
        root = pow(self.w, 1 << self.k1, self.r)
        res = [0] * (R1 * N2)
        for t in range(R1):
            row = [vals[(v // W) * R1 * W + t * W + (v % W)] for v in range(N2)]
            tr = [v * self.scale % self.r for v in O.fft_ff(row, root, self.r)]
            for b in range(N2):
                if blocked:
                    res[(b // W) * R1 * W + t * W + (b % W)] = tr[b]
                else:
                    res[t * N2 + b] = tr[b]
        put(out, res)


def _rows_twist(self, rows, row_base):
    """mode of kzg_ntt_rows_twist_device: N2-point transforms along the rows, output (rho, beta) times w^(rho beta)"""
    R, N2 = rows.shape[0], rows.shape[1]
    vals = ints_of(rows)
    root = pow(self.w, 1 << self.k1, self.r)
    out = [0] * (R * N2)
    for t in range(R):
        tr = O.fft_ff(vals[t * N2:(t + 1) * N2], root, self.r)
        for b in range(N2):
            out[t * N2 + b] = tr[b] * pow(self.w, (row_base + t) * b, self.r) % self.r * self.scale % self.r
    put(rows, out)


def _columns_plain(self, M):
    N1, W = M.shape[0], M.shape[1]
    vals = ints_of(M)
    root = pow(self.w, 1 << self.k2, self.r)
    out = [0] * (N1 * W)
    for c in range(W):
        col = O.fft_ff([vals[t * W + c] for t in range(N1)], root, self.r)
        for t in range(N1):
            out[t * W + c] = col[t]
    put(M, out)


OracleNttOps.rows_twist = _rows_twist
OracleNttOps.columns_plain = _columns_plain


class OracleShardBackend:
    min_distributed_log = 2            # the all-to-all choreography even at 16 elements

    def __init__(self, curve):
        self.cv = O.curve(curve)
        self.r = self.cv.r

    # -- vectors
    def upload(self, values):
        return tensor_of([int(v) % self.r for v in values])

    def upload_limbs(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4).view(np.int64).copy())

    def zeros(self, m):
        return torch.zeros((m, 4), dtype=torch.int64)

    def const(self, m, v):
        return tensor_of([int(v) % self.r] * m)

    def download(self, t):
        return ints_of(t)

    def _bin(self, a, b, f):
        return tensor_of([f(x, y) % self.r for x, y in zip(ints_of(a), ints_of(b))])

    def mul(self, a, b): return self._bin(a, b, lambda x, y: x * y)
    def add(self, a, b): return self._bin(a, b, lambda x, y: x + y)
    def sub(self, a, b): return self._bin(a, b, lambda x, y: x - y)

    def lincomb(self, m, terms):
        acc = [0] * m
        for s, t in terms:
            v = ints_of(t)
            for i in range(min(m, len(v))):
                acc[i] = (acc[i] + int(s) * v[i]) % self.r
        return tensor_of(acc)

    def mul_powers(self, a, s, c0=1):
        out, p = [], int(c0) % self.r
        for v in ints_of(a):
            out.append(v * p % self.r)
            p = p * int(s) % self.r
        return tensor_of(out)

    def inverse(self, a):
        return tensor_of([pow(v, -1, self.r) if v else 0 for v in ints_of(a)])

    def prefix_product(self, a):
        out, p = [], 1
        for v in ints_of(a):
            out.append(p)
            p = p * v % self.r
        return tensor_of(out)

    def eval(self, coeffs, z):
        return O.poly_eval(ints_of(coeffs), int(z) % self.r, self.r)

    def set_entries(self, t, updates):
        v = ints_of(t)
        for i, d in updates:
            v[i] = (v[i] + int(d)) % self.r
        put(t, v)

    def any_nonzero(self, t):
        return bool(t.any().item()) if t.numel() else False

    # -- transforms
    def full_ntt(self, t, w, inverse):
        v = ints_of(t)
        put(t, O.ifft_ff(v, int(w), self.r) if inverse else O.fft_ff(v, int(w), self.r))

    def ntt_ops(self, log_n, w, inverse):
        return OracleNttOps(log_n, int(w), self.r, inverse)

    # -- key shards, MSMs, openings
    def _pt(self, p):
        a = O.normalize(p, self.cv)
        return (1, 1, 0) if a is None else (a[0], a[1], 1)

    def key_shard(self, ck, start, count):
        return [O.Z1() if int(p[2]) == 0 else (int(p[0]), int(p[1]), 1) for p in ck[start:start + count]]

    def commit_begin(self, shard, tensors):
        return [self._pt(c) for c in O.commit(shard, [ints_of(t) for t in tensors], self.cv)]

    def commit_end(self, handle):
        return handle

    def open_begin(self, tensors, z, xi):
        polys = [ints_of(t) for t in tensors]
        n = max(len(p) for p in polys)
        comb = (O.combine(polys, int(xi), self.r) + [0] * n)[:n]
        self._slice = comb
        return O.poly_eval(comb, int(z), self.r)

    def open_finish(self, shard, z, carry, first):
        local, r, z = self._slice, self.r, int(z)
        ext = local + [int(carry)]
        S = [0] * (len(ext) + 1)
        for j in range(len(ext) - 1, -1, -1):
            S[j] = (ext[j] + z * S[j + 1]) % r
        vec, ev = (S[1:len(local)], S[0]) if first else (S[0:len(local)], None)
        assert len(vec) <= len(shard)
        return (self._pt(O.commit(shard[:len(vec)], [vec], self.cv)[0]) if vec else (1, 1, 0)), ev
