"""ONE PLONK proof (plonk/prover.py:24-212) with its VECTOR work split over G ranks, one GPU each -- BASELINE config 5
("all commits/NTTs on 8xMI355X") beyond dealing the MSMs (sharding.ProofSharding replicates the transforms, the
accumulator and the quotient on every rank: 18 of a 2^20-gate round's 38 ms, so at most 1.85x on eight GPUs).

Every vector of the proof lives as a shard:

  * values on H and coefficient vectors: contiguous range [g n/G, (g+1) n/G) per rank; the few coefficients a
    blinding term adds above X^n (plonk/prover.py:72-75: (b1 X + b2)(X^n - 1)) are a TAIL of scalars every rank knows
    (the blinders are shared), carried beside the shards and committed by the last rank;
  * evaluations on the coset of the 4n-point subgroup: the transposed layout of the 4n four-step
    (sharding.ShardedTransforms) -- the quotient is element-wise, it does not care about the order.

Per round (reference lines in brackets):
  1  [83-89]   three wire columns: distributed INTT (range -> range), blind, partial commitments against a key shard
               (every rank a whole local MSM over its range, G partial points gathered and added: sharding.py "range
               mode"); coset evaluations of a, b, c, PI (range -> transposed; a tail enters by linearity:
               p(x) += x^n (t0 + t1 x + ..) on the coset points);
  2  [243-264] accumulator: ratios by batch inversion, LOCAL exclusive prefix product, one field element of carry per
               rank (the product of the ranks below); INTT, blind, commit, coset evaluation; z(gX) on the coset is a
               shift by four in natural order = four rows of the next rank in the transposed layout (one small gather);
  3  [297-316] quotient element-wise on the coset shards, division by Z_H (four values), inverse 4n transform
               (transposed -> range), split into t_lo / t_mid / t_hi by a block re-partition (every rank sends at most
               four blocks of n/G), commit;
  4  [140-150] evaluations at zeta: Horner over the local range times zeta^lo, one gather of six field elements per rank;
  5  [358-414, 184-185] r(X) as a scalar combination of shards; the two openings through kzg_open_shard_begin /
               _finish (slice evaluation, one field element per rank, carry, quotient slice against a key shard that
               starts at lo - 1).
All ranks see the same points and evaluations after every exchange, hence the same challenges and the same proof.

The algebra is a BACKEND (GpuShardBackend below: the engine through DeviceAlgebra; tests inject the oracle on CPU so that
the choreography runs under gloo without a GPU).  G must be a power of two with n/G >= 8; other world sizes keep
ProofSharding's dealt MSMs (make_prover)."""
import numpy as np
import torch

from . import _native
from .kzg import KZG
from .plonk import Domain
from .sharding import (FR_BYTES, POINT_BYTES, ShardedTransforms, all_gather_bytes, all_to_all_rows, pack_point,
                       unpack_point, _rank, _world)
from .transcript import Transcript

TAIL = 6                      # coefficients above X^n a polynomial of the proof can carry (t_hi: n + 6 in all)


class SPoly:
    """coefficient vector of logical length n + len(tail): `local` = this rank's range of the first n, `tail` = the
    coefficients of X^n, X^(n+1), .. as ints (the same on every rank)"""

    def __init__(self, local, tail=()):
        self.local, self.tail = local, [int(v) for v in tail]


class GpuShardBackend:
    """The engine as the sharded prover's algebra: DeviceAlgebra's vector primitives, the device transforms, local MSMs
    against key shards and the sharded opening's two steps."""

    min_distributed_log = 13           # kzg_ntt_columns_device / _rows_exchange_device need log_n > 12

    def __init__(self, alg):
        self.alg = alg
        self.ctx = alg.ctx
        self.r = alg.r

    # -- vectors
    def upload(self, values):
        return self.alg.upload(values)

    def upload_limbs(self, arr):
        return self.alg.upload_limbs(arr)

    def zeros(self, m):
        return self.alg.zeros(m)

    def const(self, m, v):
        return self.alg.const(m, v)

    def download(self, t):
        return self.alg.download(t)

    def mul(self, a, b): return self.alg.mul(a, b)
    def add(self, a, b): return self.alg.add(a, b)
    def sub(self, a, b): return self.alg.sub(a, b)
    def lincomb(self, m, terms): return self.alg.lincomb(m, terms)
    def mul_powers(self, a, s, c0=1): return self.alg.mul_powers(a, s, c0)
    def inverse(self, a): return self.alg.inverse(a)
    def prefix_product(self, a): return self.alg.prefix_product(a)
    def eval(self, coeffs, z): return self.alg.eval(coeffs, z)
    def set_entries(self, t, updates): return self.alg.set_entries(t, updates)

    def any_nonzero(self, t):
        return bool(t.any().item()) if t.numel() else False

    # -- transforms
    def full_ntt(self, t, w, inverse):
        self.alg.ntt(t, w, inverse)

    def ntt_ops(self, log_n, w, inverse):
        from .sharding import GpuNttOps
        return GpuNttOps(self.ctx, log_n, _native.int_to_words(int(w) % self.r), inverse)

    # -- key shards, MSMs, openings
    def key_shard(self, ck, start, count):
        """points [start, start + count) of the commitment key as a device table of their own"""
        xy, inf = ck.srs.export(start, count)
        return self.ctx.srs_load_g1(np.ascontiguousarray(xy), np.ascontiguousarray(inf))

    def commit_begin(self, shard, tensors):
        alg = self.alg
        stride = max(t.shape[0] for t in tensors)
        pack = torch.stack([alg.padded(t, stride) for t in tensors]).contiguous()
        xy = np.zeros((len(tensors), 2 * self.ctx.fp_limbs), dtype=np.uint64)
        inf = np.zeros(len(tensors), dtype=np.uint8)
        self.ctx.commit_device_async(shard, pack.data_ptr(), [t.shape[0] for t in tensors], stride, xy, inf)
        return pack, xy, inf

    def commit_end(self, handle):
        _, xy, inf = handle
        self.ctx.commit_flush()
        L = self.ctx.fp_limbs
        out = []
        for row, f in zip(xy, inf):
            if f:
                out.append((1, 1, 0))
            else:
                v = _native.limbs_to_ints(row.reshape(2, L))
                out.append((v[0], v[1], 1))
        return out

    def open_begin(self, tensors, z, xi):
        alg = self.alg
        stride = max(t.shape[0] for t in tensors)
        self._open_pack = torch.stack([alg.padded(t, stride) for t in tensors]).contiguous()
        h = self.ctx.open_shard_begin(self._open_pack.data_ptr(), [t.shape[0] for t in tensors], stride,
                                      _native.int_to_words(int(z) % self.r), _native.int_to_words(int(xi) % self.r))
        return _native.limbs_to_ints(h.reshape(1, 4))[0]

    def open_finish(self, shard, z, carry, first):
        xy, inf, ev = self.ctx.open_shard_finish(shard, _native.int_to_words(int(z) % self.r),
                                                 _native.int_to_words(int(carry) % self.r), first)
        L = self.ctx.fp_limbs
        pt = (1, 1, 0) if inf[0] else tuple(_native.limbs_to_ints(xy.reshape(2, L))) + (1,)
        return pt, (_native.limbs_to_ints(ev.reshape(1, 4))[0] if first else None)


class ShardedProver:
    def __init__(self, curve_type, backend, group=None):
        self.kzg = KZG(curve_type)
        self.be = backend
        self.group = group
        self.tf = ShardedTransforms(backend.ntt_ops, backend.full_ntt, group=group,
                                    min_log=getattr(backend, "min_distributed_log", 13))
        self.exchanges = 0            # record gathers (points, field elements); the transforms count their own
        self._dom_cache = {}

    # ---- ranks
    @property
    def world(self):
        return _world(self.group)

    @property
    def rank(self):
        return _rank(self.group)

    def _gather(self, payload):
        self.exchanges += 1
        return all_gather_bytes(payload, self.group)

    def _gather_ints(self, values):
        """every rank's list of field elements: [rank][i]"""
        blobs = self._gather(b"".join(int(v).to_bytes(FR_BYTES, "little") for v in values))
        k = len(values)
        return [[int.from_bytes(b[i * FR_BYTES:(i + 1) * FR_BYTES], "little") for i in range(k)] for b in blobs]

    # ---- layout helpers
    def _t_powers(self, log_big, s, c0):
        """c0 * s^(global index) for this rank's transposed-layout shard of a 2^log_big vector: the index of element
        (t, b) is b * N1 + g * R1 + t, so the vector is the outer product of a column of R1 and a row of N2 powers"""
        be, r, G, g = self.be, self.kzg.curve_order, self.world, self.rank
        k1 = (log_big + 1) // 2
        N1, N2 = 1 << k1, 1 << (log_big - k1)
        R1 = N1 // G
        col = be.mul_powers(be.const(R1, 1), s, int(c0) * pow(int(s), g * R1, r) % r)         # over t
        row = be.mul_powers(be.const(N2, 1), pow(int(s), N1, r), 1)                            # over b
        return be.mul(col.repeat_interleave(N2, dim=0).contiguous(), row.repeat(R1, 1).contiguous())

    def _domain_constants(self, n, g):
        key = (n, int(g), self.world, self.rank)
        if key not in self._dom_cache:
            be, Fq, r, G = self.be, self.kzg.Fq, self.kzg.curve_order, self.world
            N4 = 4 * n
            log4 = N4.bit_length() - 1
            w4, K = int(Fq.root_of_unity(N4)), int(Fq.multiplicative_generator())
            m, m4 = n // G, N4 // G
            lo = self.rank * m
            ones, ones4 = be.const(m, 1), be.const(m4, 1)
            xs = self._t_powers(log4, w4, K)                                        # the coset points, transposed layout
            xn = self._t_powers(log4, pow(w4, n, r), pow(K, n, r))                  # x^n: four distinct values
            zh = be.sub(xn, ones4)
            xnx = [xn, be.mul(xn, xs)]
            xnx.append(be.mul(xnx[1], xs))                                          # x^n, x^(n+1), x^(n+2)
            self._dom_cache.clear()
            self._dom_cache[key] = {
                "ones": ones, "idH": be.mul_powers(ones, g, pow(int(g), lo, r)), "ones4": ones4, "xs": xs,
                "zh_inv": be.inverse(zh), "w4": w4, "K": K, "log4": log4, "xnx": xnx,
                "l1": be.mul(zh, be.inverse(be.lincomb(m4, [(n, xs), (-n, ones4)])))}
        return self._dom_cache[key]

    def _on_coset(self, sp, n, D):
        return self._on_coset_batch([sp], n, D)[0]

    def _on_coset_batch(self, sps, n, D):
        """evaluations of sharded polynomials on the coset K * H', |H'| = 4n, in the transposed layout; the batch
        shares one set of all-to-alls"""
        be, r = self.be, self.kzg.curve_order
        m = n // self.world
        lo = self.rank * m
        log_n = n.bit_length() - 1
        shifted = [be.mul_powers(sp.local, D["K"], pow(D["K"], lo, r)) for sp in sps]    # c_i K^i, i = lo + local index
        Es = self.tf.padded_to_T_batch(shifted, log_n, D["log4"], D["w4"])
        out = []
        for sp, E in zip(sps, Es):
            if any(sp.tail):                                                        # + x^n (t0 + t1 x + t2 x^2)
                assert len(sp.tail) <= 3
                E = be.lincomb(E.shape[0], [(1, E)] + [(tv, D["xnx"][j]) for j, tv in enumerate(sp.tail) if tv])
            out.append(E)
        return out

    def _blind(self, coeffs_local, blinders):
        """+ (b_k X^k + ..) (X^n - 1): the low coefficients live on rank 0, the high ones are the tail"""
        be = self.be
        local = coeffs_local.clone()
        if self.rank == 0:
            be.set_entries(local, [(k, -int(bk)) for k, bk in enumerate(blinders)])
        return SPoly(local, blinders)

    def _with_tail(self, sp):
        """what this rank commits / opens of a sharded polynomial: its range, and on the last rank the tail too"""
        be = self.be
        if self.rank != self.world - 1:
            return sp.local
        tail = list(sp.tail) + [0] * (TAIL - len(sp.tail))
        return torch.cat([sp.local, be.upload(tail)]).contiguous()

    def _key_shards(self, ipk, n):
        """commit shard: key points of this rank's range (+ tail); open shard: the same range moved down by one (the
        quotient's coefficient j-1 is S_j), starting at 0 on rank 0"""
        cache = ipk.setdefault("_key_shards", {})
        key = (self.world, self.rank)
        if key not in cache:
            m = n // self.world
            lo = self.rank * m
            hi = lo + m + (TAIL if self.rank == self.world - 1 else 0)
            start = 0 if self.rank == 0 else lo - 1
            cache[key] = (self.be.key_shard(ipk["ck"], lo, hi - lo), self.be.key_shard(ipk["ck"], start, hi - 1 - start))
        return cache[key]

    def _circuit(self, ipk, n, D):
        """this rank's ranges of the preprocessed polynomials and sigma values, and their coset evaluations"""
        cache = ipk.setdefault("_sharded_circuit", {})
        key = (self.world, self.rank)
        if key not in cache:
            m = n // self.world
            lo = self.rank * m
            C = {k: SPoly(v[lo:lo + m].contiguous()) for k, v in ipk["coeffs"].items()}
            S = {k: v[lo:lo + m].contiguous() for k, v in ipk["sigma_values"].items()}
            names = ("qM", "qL", "qR", "qO", "qC", "S_sigma1", "S_sigma2", "S_sigma3")
            E = dict(zip(names, self._on_coset_batch([C[k] for k in names], n, D)))
            cache[key] = (C, S, E)
        return cache[key]

    # ---- distributed pieces of the rounds
    def _commit_round(self, shard, polys):
        """queue this rank's partial commitments (its range of every polynomial, the tail on the last rank) on the
        commit pipeline; _commit_collect gathers and adds the ranks' partial points"""
        return self.be.commit_begin(shard, [self._with_tail(p) for p in polys])

    def _sum_points(self, pts):
        """the ranks' partial results added up: the library's host group law (kzg_g1_sum, one inversion per sum)"""
        try:
            return _native.g1_sum(self.kzg.curve_type, pts)
        except _native.NativeUnavailable:
            acc = self.kzg.Z1
            for p in pts:
                acc = self.kzg.add(acc, p)
            return acc

    def _commit_collect(self, handle, k):
        parts = self.be.commit_end(handle)
        blobs = self._gather(b"".join(pack_point(p) for p in parts))
        return [self._sum_points([unpack_point(blob[i * POINT_BYTES:(i + 1) * POINT_BYTES]) for blob in blobs])
                for i in range(k)]

    def _evals(self, pairs, n):
        """[(sharded polynomial, point)] -> values: Horner over the local range times z^lo, tails on the host, one gather"""
        be, r = self.be, self.kzg.curve_order
        m = n // self.world
        lo = self.rank * m
        mine = [be.eval(sp.local, z) * pow(int(z), lo, r) % r for sp, z in pairs]
        tot = [0] * len(pairs)
        for row in self._gather_ints(mine):
            tot = [(a + b) % r for a, b in zip(tot, row)]
        return [(t + sum(tv * pow(int(z), n + j, r) for j, tv in enumerate(sp.tail))) % r
                for t, (sp, z) in zip(tot, pairs)]

    def _prefix_product(self, a):
        """exclusive prefix product over the WHOLE vector: local scan, then the product of the ranks below as carry"""
        be, r = self.be, self.kzg.curve_order
        local = be.prefix_product(a)
        last = be.download(torch.cat([local[-1:], a[-1:]]))
        totals = [row[0] for row in self._gather_ints([last[0] * last[1] % r])]
        carry = 1
        for h in range(self.rank):
            carry = carry * totals[h] % r
        return be.mul_powers(local, 1, carry)

    def _split_quotient(self, t_nat, n):
        """t's coefficients in range order over 4n (rank s: [4 s m, 4 (s+1) m), m = n/G) -> t_lo, t_mid, t_hi as range
        shards over n: block beta = q G + d (q-th polynomial, rank d's range) sits on rank beta // 4.  Returns the three
        shards, the six coefficients above X^(3n) and whether anything non-zero lies beyond them."""
        be, G, g = self.be, self.world, self.rank
        m = n // G
        # equal-sized slots (cap blocks per destination; unused ones travel as they are): an all-to-all with uniform
        # splits is the collective every backend implements the same way
        cap = max(1, 4 // G)
        send = torch.zeros((G, cap, m, 4), dtype=t_nat.dtype, device=t_nat.device)
        fill = [0] * G
        for j in range(4):
            beta = 4 * g + j
            if beta < 3 * G:
                d = beta % G
                send[d, fill[d]] = t_nat[j * m:(j + 1) * m]
                fill[d] += 1
        self.tf.exchanges += 1
        got = all_to_all_rows(send.view(G * cap * m, 4), [cap * m] * G, [cap * m] * G, self.group).view(G, cap, m, 4)
        parts = {}
        for s in range(G):
            slot = 0
            for beta in range(4 * s, 4 * s + 4):
                if beta < 3 * G and beta % G == g:
                    parts[beta // G] = got[s, slot]
                    slot += 1
        assert sorted(parts) == [0, 1, 2]
        # the tail [3n, 3n + 6) and the remainder check: block 3 G lives on rank (3 G) // 4
        owner = (3 * G) // 4
        off = (3 * G - 4 * owner) * m
        tail, dirty = [0] * TAIL, False
        if g == owner:
            tail = be.download(t_nat[off:off + TAIL])
            dirty = be.any_nonzero(t_nat[off + TAIL:])
        elif g > owner:
            dirty = be.any_nonzero(t_nat)
        rows = self._gather_ints(tail + [1 if dirty else 0])
        return parts[0], parts[1], parts[2], rows[owner][:TAIL], any(row[TAIL] for row in rows)

    def _open(self, oshard, polys, z, xi, n):
        """kzg.py:122-159 on range shards: slice evaluations, carries, partial proofs (two gathers)"""
        be, r, G, g = self.be, self.kzg.curve_order, self.world, self.rank
        m = n // G
        H = [row[0] for row in self._gather_ints([be.open_begin([self._with_tail(p) for p in polys], z, xi)])]
        hi = (g + 1) * m + (TAIL if g == G - 1 else 0)
        carry = sum(H[h] * pow(int(z), h * m - hi, r) for h in range(g + 1, G)) % r
        part, _ = be.open_finish(oshard, z, carry, g == 0)
        return self._sum_points([unpack_point(blob) for blob in self._gather(pack_point(part))])

    # ---- the proof
    def prove(self, ipk, x, w, blinders=None, trace=None):
        try:
            return self._prove(ipk, x, w, blinders, trace)
        except BaseException:
            flush = getattr(getattr(self.be, "ctx", None), "commit_flush", None)
            if flush:
                try:
                    flush()
                except Exception:   # noqa: BLE001
                    pass
            raise

    def _prove(self, ipk, x, w, blinders, trace):
        kzg, Fq, be = self.kzg, self.kzg.Fq, self.be
        r, G, g_rank = kzg.curve_order, self.world, self.rank
        sub = ipk["subgroups"]
        n, g, k1, k2 = sub["n"], sub["g"], int(sub["k1"]), int(sub["k2"])
        if G & (G - 1) or n % G or n // G < 8:
            raise ValueError("the vector-sharded prover needs a power-of-two world size with n / G >= 8")
        m = n // G
        lo = g_rank * m
        log_n = n.bit_length() - 1
        dom = Domain.__new__(Domain)
        dom.Fq, dom.n, dom.g = Fq, n, g
        tr = Transcript("plonk-proof", Fq)
        tr.append_message("public-inputs", x)
        x_ints = [int(v) % r for v in x]
        w_limbs = (np.ascontiguousarray(w, dtype=np.uint64).reshape(-1, 4) if isinstance(w, np.ndarray)
                   else _native.ints_to_limbs([int(v) % r for v in w]).reshape(-1, 4))
        x_limbs = _native.ints_to_limbs(x_ints).reshape(-1, 4)
        nx = x_limbs.shape[0]
        assert nx + w_limbs.shape[0] == 3 * n

        def column(i):
            """rows [i n + lo, i n + lo + m) of x ++ w: this rank's range of wire column i"""
            a, b = i * n + lo, i * n + lo + m
            parts = []
            if a < nx:
                parts.append(x_limbs[a:min(b, nx)])
            if b > nx:
                parts.append(w_limbs[max(a, nx) - nx:b - nx])
            return be.upload_limbs(np.concatenate(parts))

        b = [int(Fq.random_element()) for _ in range(11)] if blinders is None else [int(v) % r for v in blinders]
        assert len(b) == 11
        b = self._gather_ints(b)[0]                                                  # drawn once, by rank 0
        D = self._domain_constants(n, g)
        ones, idH, ones4, xs = D["ones"], D["idH"], D["ones4"], D["xs"]
        cshard, oshard = self._key_shards(ipk, n)
        C, S, E_circ = self._circuit(ipk, n, D)
        N4 = 4 * n
        m4 = N4 // G

        # round 1
        vals = [column(i) for i in range(3)]
        pi_vals = be.zeros(m)
        if lo < nx:
            cnt = min(nx, lo + m) - lo
            pi_vals[:cnt] = be.upload([(-v) % r for v in x_ints[lo:lo + cnt]])
        co = self.tf.natural_batch(vals + [pi_vals], log_n, g, True)                # four INTTs, one set of all-to-alls
        wires = [self._blind(co[i], [b[2 * i + 1], b[2 * i]]) for i in range(3)]
        a_c, b_c, c_c = wires
        h1 = self._commit_round(cshard, wires)
        PI_c = SPoly(co[3])
        E = dict(zip(("a", "b", "c", "PI"), self._on_coset_batch([a_c, b_c, c_c, PI_c], n, D)))
        E.update(E_circ)
        gate = be.add(be.add(be.mul(be.mul(E["a"], E["b"]), E["qM"]), be.mul(E["a"], E["qL"])),
                      be.add(be.mul(E["b"], E["qR"]), be.mul(E["c"], E["qO"])))
        gate = be.add(gate, be.add(E["PI"], E["qC"]))
        wire_comms = self._commit_collect(h1, 3)
        tr.append_message("round1-commitments", wire_comms)
        beta, gamma = int(tr.get_challenge("beta")), int(tr.get_challenge("gamma"))

        # round 2
        num = den = None
        for v, shift, sig in ((vals[0], 1, S["S_sigma1"]), (vals[1], k1, S["S_sigma2"]), (vals[2], k2, S["S_sigma3"])):
            fn = be.lincomb(m, [(1, v), (beta * shift, idH), (gamma, ones)])
            fd = be.lincomb(m, [(1, v), (beta, sig), (gamma, ones)])
            num = fn if num is None else be.mul(num, fn)
            den = fd if den is None else be.mul(den, fd)
        z_vals = self._prefix_product(be.mul(num, be.inverse(den)))
        z_c = self._blind(self.tf.natural(z_vals, log_n, g, True), [b[8], b[7], b[6]])
        h2 = self._commit_round(cshard, [z_c])
        E["z"] = self._on_coset(z_c, n, D)
        zw = self.tf.T_shift(E["z"], D["log4"], 4)                                   # z(g x): g = w4^4
        p1 = p2 = None
        for key, shift, sig in (("a", 1, "S_sigma1"), ("b", k1, "S_sigma2"), ("c", k2, "S_sigma3")):
            f1 = be.lincomb(m4, [(1, E[key]), (beta * shift, xs), (gamma, ones4)])
            f2 = be.lincomb(m4, [(1, E[key]), (beta, E[sig]), (gamma, ones4)])
            p1 = f1 if p1 is None else be.mul(p1, f1)
            p2 = f2 if p2 is None else be.mul(p2, f2)
        perm = be.sub(be.mul(p1, E["z"]), be.mul(p2, zw))
        l1t = be.mul(be.sub(E["z"], ones4), D["l1"])
        z_comm = self._commit_collect(h2, 1)[0]
        tr.append_message("round2-commitment", z_comm)
        alpha = int(tr.get_challenge("alpha"))

        # round 3
        numer = be.lincomb(m4, [(1, gate), (alpha, perm), (alpha * alpha, l1t)])
        t_ev = be.mul(numer, D["zh_inv"])
        t_nat = self.tf.T_to_natural(t_ev, D["log4"], D["w4"], True)                 # range order over 4n
        Kinv = pow(D["K"], -1, r)
        t_nat = be.mul_powers(t_nat, Kinv, pow(Kinv, g_rank * m4, r))               # back from the coset
        lo_l, mid_l, hi_l, t_tail, dirty = self._split_quotient(t_nat, n)
        assert not dirty, "constraint system is not satisfied (quotient has a remainder)"
        t_lo, t_mid, t_hi = SPoly(lo_l.clone(), [b[9]]), SPoly(mid_l.clone(), [b[10]]), SPoly(hi_l.clone(), t_tail)
        if g_rank == 0:
            be.set_entries(t_mid.local, [(0, -b[9])])
            be.set_entries(t_hi.local, [(0, -b[10])])
        t_comms = self._commit_collect(self._commit_round(cshard, [t_lo, t_mid, t_hi]), 3)
        tr.append_message("round3-commitments", t_comms)
        zeta = int(tr.get_challenge("zeta"))

        # round 4
        zg = zeta * int(g) % r
        names = ("a", "b", "c", "s_sigma1", "s_sigma2", "z_omega")
        vals4 = self._evals([(a_c, zeta), (b_c, zeta), (c_c, zeta), (C["S_sigma1"], zeta), (C["S_sigma2"], zeta),
                             (z_c, zg), (PI_c, zeta)], n)
        ev = dict(zip(names, vals4[:6]))
        PIz = vals4[6]
        evF = {k: Fq(v) for k, v in ev.items()}
        tr.append_message("round4-evaluations", [evF[k] for k in names])
        v = int(tr.get_challenge("v"))

        # round 5
        za, zb, zc, s1, s2, zo = (ev[k] for k in names)
        zn = pow(zeta, n, r)
        L1z = int(dom.lagrange_1_at(Fq(zeta)))
        f1 = (za + beta * zeta + gamma) * (zb + beta * k1 * zeta + gamma) * (zc + beta * k2 * zeta + gamma) % r
        f2 = (za + beta * s1 + gamma) * (zb + beta * s2 + gamma) * zo % r
        const = (PIz - alpha * f2 * (zc + gamma) - alpha * alpha * L1z) % r
        terms = [(za * zb, C["qM"]), (za, C["qL"]), (zb, C["qR"]), (zc, C["qO"]), (1, C["qC"]),
                 (alpha * f1 + alpha * alpha * L1z, z_c), (-alpha * f2 * beta, C["S_sigma3"]),
                 (-(zn - 1), t_lo), (-(zn - 1) * zn, t_mid), (-(zn - 1) * zn * zn, t_hi)]
        r_local = be.lincomb(m, [(s, p.local) for s, p in terms])
        if g_rank == 0:
            be.set_entries(r_local, [(0, const)])
        r_tail = [sum(s * (p.tail[j] if j < len(p.tail) else 0) for s, p in terms) % r for j in range(TAIL)]
        r_c = SPoly(r_local, r_tail)
        assert self._evals([(r_c, zeta)], n)[0] == 0, "r(zeta) should be zero"       # plonk/prover.py:171
        if trace is not None:
            trace.update(beta=beta, gamma=gamma, alpha=alpha, zeta=zeta, v=v, evaluations=dict(ev),
                         a=a_c, b=b_c, c=c_c, z=z_c, PI=PI_c, t_lo=t_lo, t_mid=t_mid, t_hi=t_hi, r=r_c)
        W_z = self._open(oshard, [r_c, a_c, b_c, c_c, C["S_sigma1"], C["S_sigma2"]], zeta, v, n)
        W_zw = self._open(oshard, [z_c], zg, v, n)
        return {"commitments": dict(zip(("a", "b", "c"), wire_comms), z=z_comm,
                                    t_lo=t_comms[0], t_mid=t_comms[1], t_hi=t_comms[2]),
                "evaluations": evF,
                "kzg_proofs": {"W_z": W_z, "W_zw": W_zw}}


def make_prover(curve_type, alg, sharding):
    """The prover for a sharding.ProofSharding: vectors split over the ranks when it asks for that and the world size
    allows it (a power of two), else the MSMs dealt over replicated vectors (plonk_device.DeviceProver)."""
    from . import plonk_device
    world = sharding.world if sharding is not None else 1
    if sharding is not None and getattr(sharding, "shard_vectors", False) and world & (world - 1) == 0:
        return ShardedProver(curve_type, GpuShardBackend(alg), group=sharding.group)
    return plonk_device.DeviceProver(curve_type, alg=alg, sharding=sharding)
