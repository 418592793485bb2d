"""Drop-in for the reference's kzg.py `KZG` class: same constructor, attributes and
method signatures (kzg.py:18-288), with the data-parallel work -- SRS generation,
commit and open -- executed by the gfx950 engine through the C ABI.

    KZG(curve_type="bn254")                          kzg.py:18
      .G1 .G2 .Z1 .Z2 .multiply .add .neg .eq .pairing .curve_order .Fq .R .X     kzg.py:40-54
    setup(max_degree) -> (ck, tau_G2)                kzg.py:56
    commit(ck, polynomials) -> [point]               kzg.py:80
    open(ck, polynomials, z, xi) -> point            kzg.py:122
    check(rk, commitments, z, evaluations, proof, xi) -> bool            kzg.py:161
    batch_check(rk, commitments_list, z_list, evaluations_list, proof_list, xi_list, r=None)   kzg.py:213

Points are py_ecc-shaped 3-tuples, always normalised: (x, y, 1), infinity (1, 1, 0).
The reference returns un-normalised projective triples whose representative
depends on py_ecc's operation order; equality there is `eq`, here plain `==` works
too.  There is no CPU fallback for commit/open/setup: they raise
_native.NativeUnavailable without the shared library and a GPU."""
from collections.abc import Sequence

import numpy as np

from . import _native
from . import curve as _curve
from .field import GF, Polynomial, PolynomialRing


class CommitmentKey(Sequence):
    """The `ck` of the reference: a sequence of G1 points [tau^i G1] (kzg.py:70-72).

    Lives on the device (kzg_srs handle) in the engine's table layout; indexing
    and iteration export points lazily so code that only passes `ck` around and
    takes len(ck) (every in-tree caller) never materialises 2^20 tuples."""

    def __init__(self, ctx, srs):
        self._ctx = ctx
        self.srs = srs
        self._cache = {}

    def __len__(self):
        return self.srs.n

    def _point(self, xy_row, inf):
        if inf:
            return (1, 1, 0)
        L = self._ctx.fp_limbs
        v = _native.limbs_to_ints(xy_row.reshape(2, L))
        return (v[0], v[1], 1)

    def __getitem__(self, i):
        if isinstance(i, slice):
            idx = range(*i.indices(len(self)))
            return [self[j] for j in idx]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("commitment key index out of range")
        if i not in self._cache:
            start = (i // 1024) * 1024
            count = min(1024, n - start)
            xy, inf = self.srs.export(start, count)
            for j in range(count):
                self._cache[start + j] = self._point(xy[j], inf[j])
        return self._cache[i]


class KZG:
    def __init__(self, curve_type="bn254"):
        if curve_type not in _curve.CURVES:
            raise ValueError(f"Unsupported curve type: {curve_type}")          # kzg.py:37
        self.curve_type = curve_type
        cv = _curve.CURVES[curve_type]
        self._cv = cv
        self._g1 = _curve.g1_group(cv)
        self._g2 = _curve.g2_group(cv)
        # curve operations with py_ecc's names (kzg.py:40-49)
        self.G1 = (cv.g1[0], cv.g1[1], 1)
        self.G2 = (cv.g2[0], cv.g2[1], (1, 0))
        self.Z1 = self._g1.Z
        self.Z2 = self._g2.Z
        self.curve_order = cv.r
        # field and polynomial ring (kzg.py:52-54)
        self.Fq = GF(cv.r)
        self.R = PolynomialRing(self.Fq, "X")
        self.X = self.R.gen()
        self._ctx = None
        self._loaded = {}          # id(list ck) -> (CommitmentKey, fingerprint); at most _KEY_CACHE entries, LRU

    # ---- py_ecc-shaped single-point operations (host; used by the verifiers) ------
    def _grp(self, pt):
        return self._g2 if isinstance(pt[0], tuple) else self._g1

    def multiply(self, pt, n):
        return self._grp(pt).multiply(pt, int(n) % self.curve_order if int(n) >= self.curve_order else int(n))

    def add(self, p1, p2):
        return self._grp(p1).add(p1, p2)

    def neg(self, pt):
        return self._grp(pt).neg(pt)

    def eq(self, p1, p2):
        return self._grp(p1).eq(p1, p2)

    def pairing(self, Q, P):
        """pairing(G2_point, G1_point) as used at kzg.py:208-209, 285-286."""
        from .pairing import pairing as _pairing
        return _pairing(Q, P, self._cv)

    # ---- engine plumbing ------------------------------------------------------------
    def _context(self):
        if self._ctx is None:
            self._ctx = _native.get_context(self.curve_type)
        return self._ctx

    _KEY_CACHE = 2                 # device tables kept for list-form keys (a 2^20-point table is 1.7 GiB)

    @staticmethod
    def _fingerprint(ck):
        """Length plus up to 16 sampled points: guards the id()-keyed cache against a recycled id
        and against in-place edits of the list (not a hash of every point: that would cost as much
        as the upload the cache avoids)."""
        n = len(ck)
        if n == 0:
            return (0,)
        step = max(1, n // 15)
        return (n,) + tuple(tuple(int(c) for c in ck[i]) for i in sorted({*range(0, n, step), n - 1}))

    def _key(self, ck):
        """Device handle for a commitment key given as CommitmentKey or as a plain
        sequence of point tuples (what the reference's callers hold)."""
        if isinstance(ck, CommitmentKey):
            return ck
        fp = self._fingerprint(ck)
        hit = self._loaded.get(id(ck))
        if hit and hit[1] == fp:
            self._loaded[id(ck)] = self._loaded.pop(id(ck))       # most recently used last
            return hit[0]
        ctx = self._context()
        L = ctx.fp_limbs
        n = len(ck)
        coords, inf = [], np.zeros(n, dtype=np.uint8)
        for i, pt in enumerate(ck):
            x, y, z = self._g1.normalize(tuple(int(c) for c in pt))
            if z == 0:
                inf[i] = 1
                x = y = 0
            coords.append(x)
            coords.append(y)
        xy = _native.ints_to_limbs(coords, L).reshape(n, 2 * L)
        key = CommitmentKey(ctx, ctx.srs_load_g1(np.ascontiguousarray(xy), inf))
        self._loaded.pop(id(ck), None)
        while len(self._loaded) >= self._KEY_CACHE:                # evict the least recently used table
            _, (old, _) = next(iter(self._loaded.items()))
            del self._loaded[next(iter(self._loaded))]
            old.srs.close()
        self._loaded[id(ck)] = (key, fp)
        return key

    def _coeffs(self, poly):
        """Coefficient ints of a polynomial given as a list, our Polynomial, or any
        object with .list() (a Sage polynomial); trailing zeros dropped like R(poly)."""
        r = self.curve_order
        if isinstance(poly, Polynomial):
            return poly.c
        if isinstance(poly, np.ndarray) and poly.dtype == np.uint64 and poly.ndim == 2 and poly.shape[1] == 4:
            # buffer fast path (as in fft_ff): canonical little-endian limbs, no per-element Python objects;
            # trailing zero coefficients dropped like everywhere else
            if poly.shape[0] and poly[-1].any():                  # the usual case: nothing to trim, nothing to scan
                return np.ascontiguousarray(poly)
            nz = np.flatnonzero(poly.any(axis=1))
            return np.ascontiguousarray(poly[:int(nz[-1]) + 1 if nz.size else 0])
        if isinstance(poly, (list, tuple)):
            c = [int(x) % r for x in poly]
        else:
            c = [int(x) % r for x in poly.list()]
        while c and c[-1] == 0:
            c.pop()
        return c

    def _pack(self, coeff_lists):
        stride = max((len(c) for c in coeff_lists), default=0)
        stride = max(stride, 1)
        if len(coeff_lists) == 1 and isinstance(coeff_lists[0], np.ndarray) and len(coeff_lists[0]):
            return coeff_lists[0].reshape(1, stride, 4), [stride], stride          # one buffer: handed over as it is
        arr = np.zeros((len(coeff_lists), stride, 4), dtype=np.uint64)
        for i, c in enumerate(coeff_lists):
            if len(c):
                arr[i, :len(c)] = c if isinstance(c, np.ndarray) else _native.ints_to_limbs(c)
        return arr, [len(c) for c in coeff_lists], stride

    def _points(self, xy, inf):
        L = self._context().fp_limbs
        out = []
        for row, f in zip(np.atleast_2d(xy), np.atleast_1d(inf)):
            if f:
                out.append(self.Z1)
            else:
                v = _native.limbs_to_ints(row.reshape(2, L))
                out.append((v[0], v[1], 1))
        return out

    # ---- the scheme -----------------------------------------------------------------
    def setup(self, max_degree, tau=None):
        """kzg.py:56-78.  `tau` may be supplied for reproducible tests (the reference
        samples it at :67 and discards it; so do we when it is not given)."""
        if tau is None:
            tau = self.Fq.random_element()
        tau = int(tau) % self.curve_order
        ctx = self._context()
        srs = ctx.srs_generate(_native.int_to_words(tau), int(max_degree) + 1)      # kzg.py:69-72
        tau_G2 = self.multiply(self.G2, tau)                                         # kzg.py:75
        return (CommitmentKey(ctx, srs), tau_G2)

    # ---- on-disk commitment key: build the SRS once, reload it later -----------
    #   bytes 0..7   magic b"KZGSRS1\0"      bytes 8..11  curve id (u32 LE: 0 bn254, 1 bls12_381)
    #   bytes 12..15 uint64 limbs per coordinate (u32 LE)   bytes 16..23 number of points n (u64 LE)
    #   then n * 2 * limbs little-endian uint64 (affine x | y, canonical), then n infinity-flag bytes
    _MAGIC = b"KZGSRS1\0"

    def save_key(self, ck, path, chunk=1 << 16):
        key = self._key(ck)
        ctx = self._context()
        n, L = len(key), ctx.fp_limbs
        with open(path, "wb") as f:
            f.write(self._MAGIC)
            f.write(int(ctx.curve_id).to_bytes(4, "little") + int(L).to_bytes(4, "little") + int(n).to_bytes(8, "little"))
            flags = []
            for start in range(0, n, chunk):
                xy, inf = key.srs.export(start, min(chunk, n - start))
                f.write(np.ascontiguousarray(xy, dtype="<u8").tobytes())
                flags.append(inf)
            f.write(np.concatenate(flags).astype(np.uint8).tobytes())

    def load_key(self, path):
        ctx = self._context()
        with open(path, "rb") as f:
            head = f.read(24)
            if head[:8] != self._MAGIC:
                raise ValueError("not a KZG SRS file")
            cid, L, n = (int.from_bytes(head[8:12], "little"), int.from_bytes(head[12:16], "little"),
                         int.from_bytes(head[16:24], "little"))
            if cid != ctx.curve_id or L != ctx.fp_limbs:
                raise ValueError("SRS file belongs to another curve")
            xy = np.frombuffer(f.read(n * 2 * L * 8), dtype="<u8").reshape(n, 2 * L).astype(np.uint64)
            inf = np.frombuffer(f.read(n), dtype=np.uint8).copy()
        return CommitmentKey(ctx, ctx.srs_load_g1(np.ascontiguousarray(xy), inf))

    def commit(self, ck, polynomials):
        """kzg.py:80-120."""
        key = self._key(ck)
        max_degree = len(key) - 1
        coeffs = [self._coeffs(p) for p in polynomials]
        for c in coeffs:
            if len(c) - 1 > max_degree:
                raise ValueError(
                    f"Polynomial degree {len(c) - 1} exceeds maximum allowed degree {max_degree}")   # kzg.py:103-106
        if not coeffs:
            return []
        arr, lens, stride = self._pack(coeffs)
        xy, inf = self._context().commit(key.srs, arr, lens, stride)
        return self._points(xy, inf)

    def open(self, ck, polynomials, z, xi):
        """kzg.py:122-159."""
        key = self._key(ck)
        coeffs = [self._coeffs(p) for p in polynomials]
        z = int(self.Fq(z))                                                           # kzg.py:144
        xi = int(self.Fq(xi))                                                         # kzg.py:145
        arr, lens, stride = self._pack(coeffs)
        try:
            xy, inf, _ = self._context().open(key.srs, arr, lens, stride, _native.int_to_words(z),
                                              _native.int_to_words(xi))
        except _native.NativeError as e:
            if e.code == _native.KZG_ERR_DEGREE:
                raise ValueError(
                    f"Polynomial degree exceeds maximum allowed degree {len(key) - 1}") from e   # via kzg.py:157 -> :103
            raise
        return self._points(xy, inf)[0]

    # ---- verification (host).  Both checks rest on one folded claim per opening:
    #      F = sum_i xi^(i+1) (C_i - v_i G1) commits to sum_i xi^(i+1) (p_i - v_i), which (X - z) divides iff the v_i are
    #      the values at z; the proof pi commits to the quotient, so  e(F, G2) = e(pi, (tau - z) G2).
    def _sum_g1(self, points):
        try:
            return _native.g1_sum(self.curve_type, points)            # the library's host group law
        except _native.NativeUnavailable:
            total = self.Z1
            for pt in points:
                total = self.add(total, pt)
            return total

    def _folded_claim(self, commitments, evaluations, xi):
        """F above.  Weights are xi^(i+1) by position in EACH list, as the reference's two loops have it
        (kzg.py:187-194; its batch form indexes the evaluations by the commitments' positions, kzg.py:262-266)."""
        weight, parts = xi, []
        for C in commitments:
            parts.append(self.multiply(C, int(weight)))
            weight *= xi
        weight, value = xi, self.Fq(0)
        for v in evaluations:
            value += weight * self.Fq(v)
            weight *= xi
        parts.append(self.multiply(self.G1, int(-value)))
        return self._sum_g1(parts)

    def check(self, rk, commitments, z, evaluations, proof, xi):
        """kzg.py:161-211: two pairings, e(F, G2) against e(pi, tau G2 - z G2)."""
        z, xi = self.Fq(z), self.Fq(xi)
        F = self._folded_claim(commitments, evaluations, xi)
        shifted_key = self.add(rk, self.neg(self.multiply(self.G2, int(z))))
        return self.pairing(self.G2, F) == self.pairing(shifted_key, proof)

    def batch_check(self, rk, commitments_list, z_list, evaluations_list, proof_list, xi_list, r=None):
        """kzg.py:213-288: the openings' equations e(F_i + z_i pi_i, G2) = e(pi_i, tau G2) added up with weights
        rho^(i+1) (rho = `r`, sampled when not given: kzg.py:244-245) -- two pairings for any number of openings."""
        rho = self.Fq(self.Fq.random_element() if r is None else r)
        weight, lhs, rhs = rho, [], []
        for commitments, z, evaluations, proof, xi in zip(commitments_list, z_list, evaluations_list, proof_list,
                                                          xi_list):
            evaluations = list(evaluations)
            if len(evaluations) < len(commitments):
                raise IndexError("list index out of range")              # kzg.py:266 reads evaluations[j] per commitment
            F = self._folded_claim(commitments, evaluations[:len(commitments)], self.Fq(xi))
            lhs.append(self.multiply(self.add(F, self.multiply(proof, int(self.Fq(z)))), int(weight)))
            rhs.append(self.multiply(proof, int(weight)))
            weight *= rho
        return self.pairing(self.G2, self._sum_g1(lhs)) == self.pairing(rk, self._sum_g1(rhs))
