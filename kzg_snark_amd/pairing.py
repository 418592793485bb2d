"""Host-side optimal-ate pairing for BN254 and BLS12-381 -- what the reference gets from
py_ecc's `pairing` (bound at kzg.py:27-35, called as pairing(G2_point, G1_point) at
kzg.py:208-209 and :285-286).  Verification is O(1): two pairings per check /
batch_check, not data-parallel, so it stays on the host (SURVEY.md section 2 row 4,
section 8f N3).

Fp12 is the polynomial ring Fp[w]/(w^12 - a*w^6 - b) (BN254: w^6 = 9 + i;
BLS12-381: w^6 = 1 + u), elements are 12-coefficient lists; G2 points are
untwisted into E(Fp12) and the Miller loop runs with affine line functions.
Written from the textbook definitions; results are only ever compared with each
other (GT equality), never with py_ecc's representation."""

_BN_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
_BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


class _Fp12:
    """Arithmetic in Fp[w]/(w^12 - c6*w^6 - c0): w^12 = c6*w^6 + c0."""

    def __init__(self, p, c6, c0):
        self.p, self.c6, self.c0 = p, c6 % p, c0 % p
        self.one = [1] + [0] * 11
        self.zero = [0] * 12

    def add(self, a, b): return [(x + y) % self.p for x, y in zip(a, b)]
    def sub(self, a, b): return [(x - y) % self.p for x, y in zip(a, b)]
    def neg(self, a): return [(-x) % self.p for x in a]
    def scalar(self, a, k): return [x * k % self.p for x in a]

    def mul(self, a, b):
        p = self.p
        t = [0] * 23
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    t[i + j] += x * y
        for k in range(22, 11, -1):          # w^k = w^(k-12) * (c6 w^6 + c0)
            v = t[k] % p
            if v:
                t[k - 6] += v * self.c6
                t[k - 12] += v * self.c0
        return [x % p for x in t[:12]]

    def sqr(self, a): return self.mul(a, a)

    def conj(self, a):
        """The p^6-power Frobenius: w -> -w."""
        return [x if i % 2 == 0 else (-x) % self.p for i, x in enumerate(a)]

    def inv(self, a):
        """Extended Euclid in Fp[w] against the modulus polynomial."""
        p = self.p
        mod = [(-self.c0) % p] + [0] * 5 + [(-self.c6) % p] + [0] * 5 + [1]

        def deg(x):
            d = len(x) - 1
            while d >= 0 and x[d] % p == 0:
                d -= 1
            return d

        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = list(a) + [0], mod
        while deg(low) > 0:
            dl, dh = deg(low), deg(high)
            # r = high divided by low
            r = [0] * 13
            temp = list(high)
            inv_lead = pow(low[dl], -1, p)
            for i in range(dh - dl, -1, -1):
                q = temp[dl + i] * inv_lead % p
                r[i] = q
                if q:
                    for c in range(dl + 1):
                        temp[c + i] = (temp[c + i] - q * low[c]) % p
            nm, new = list(hm), list(high)
            for i in range(13):
                ri = r[i]
                if ri:
                    for j in range(13 - i):
                        nm[i + j] = (nm[i + j] - lm[j] * ri) % p
                        new[i + j] = (new[i + j] - low[j] * ri) % p
            lm, low, hm, high = nm, new, lm, low
        c = pow(low[0], -1, p)
        return [x * c % p for x in lm[:12]]

    def pow(self, a, e):
        r = self.one
        for bit in bin(e)[2:]:
            r = self.sqr(r)
            if bit == "1":
                r = self.mul(r, a)
        return r

    def eq(self, a, b): return [x % self.p for x in a] == [x % self.p for x in b]


class _PairingCurve:
    def __init__(self, name, p, r, c6, c0, shift, loop, bn_extra, untwist_divides):
        self.name, self.p, self.r = name, p, r
        self.K = _Fp12(p, c6, c0)
        self.shift = shift                    # Fp2 embedding: a0 + a1*i -> (a0 - shift*a1) + a1*w^6
        self.loop = loop
        self.bn_extra = bn_extra
        self.untwist_divides = untwist_divides
        K = self.K
        w = [0, 1] + [0] * 10
        self.w2 = K.mul(w, w)
        self.w3 = K.mul(self.w2, w)
        if untwist_divides:
            self.w2, self.w3 = K.inv(self.w2), K.inv(self.w3)
        self.final_exp = (p ** 6 + 1) // r    # hard part after the easy (p^6 - 1) step

    def embed_fp2(self, a):
        out = [0] * 12
        out[0] = (a[0] - self.shift * a[1]) % self.p
        out[6] = a[1] % self.p
        return out

    def untwist(self, Q):
        """E'(Fp2) -> E(Fp12)."""
        K = self.K
        return (K.mul(self.embed_fp2(Q[0]), self.w2), K.mul(self.embed_fp2(Q[1]), self.w3))

    def cast_g1(self, P):
        return ([P[0] % self.p] + [0] * 11, [P[1] % self.p] + [0] * 11)

    # affine arithmetic on E(Fp12): y^2 = x^3 + b
    def _double(self, R):
        K = self.K
        x, y = R
        m = K.mul(K.scalar(K.sqr(x), 3), K.inv(K.scalar(y, 2)))
        nx = K.sub(K.sqr(m), K.scalar(x, 2))
        ny = K.sub(K.mul(m, K.sub(x, nx)), y)
        return (nx, ny), m

    def _add(self, R, Q):
        K = self.K
        m = K.mul(K.sub(Q[1], R[1]), K.inv(K.sub(Q[0], R[0])))
        nx = K.sub(K.sub(K.sqr(m), R[0]), Q[0])
        ny = K.sub(K.mul(m, K.sub(R[0], nx)), R[1])
        return (nx, ny), m

    def _line(self, m, R, P):
        """Value at P of the line of slope m through R."""
        K = self.K
        return K.sub(K.mul(m, K.sub(P[0], R[0])), K.sub(P[1], R[1]))

    def miller(self, Q, P):
        K = self.K
        R, f = Q, K.one
        for bit in bin(self.loop)[3:]:
            nR, m = self._double(R)
            f = K.mul(K.sqr(f), self._line(m, R, P))
            R = nR
            if bit == "1":
                if K.eq(R[0], Q[0]):          # vertical line (does not occur for order-r inputs)
                    f = K.mul(f, K.sub(P[0], R[0]))
                    R = None
                else:
                    nR, m = self._add(R, Q)
                    f = K.mul(f, self._line(m, R, P))
                    R = nR
        if self.bn_extra:
            p = self.p
            Q1 = (K.pow(Q[0], p), K.pow(Q[1], p))
            nQ2 = (K.pow(Q1[0], p), K.neg(K.pow(Q1[1], p)))
            nR, m = self._add(R, Q1)
            f = K.mul(f, self._line(m, R, P))
            R = nR
            _, m = self._add(R, nQ2)
            f = K.mul(f, self._line(m, R, P))
        return f

    def final_exponentiate(self, f):
        K = self.K
        f = K.mul(K.conj(f), K.inv(f))         # f^(p^6 - 1)
        return K.pow(f, self.final_exp)        # ^((p^6 + 1)/r)


_BN = _PairingCurve(
    "bn254", _BN_P, 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    c6=18, c0=-82, shift=9, loop=29793968203157093288, bn_extra=True, untwist_divides=False)
_BLS = _PairingCurve(
    "bls12_381", _BLS_P, 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    c6=2, c0=-2, shift=1, loop=15132376222941642752, bn_extra=False, untwist_divides=True)
_CURVES = {"bn254": _BN, "bls12_381": _BLS}


def pairing(Q, P, cv):
    """e(P, Q) for Q in G2 (Fp2-coordinate 3-tuple), P in G1 (int 3-tuple), both normalised
    (z = 1 / (1, 0)) or infinity (z = 0).  Returns a tuple (hashable, comparable with ==)."""
    pc = _CURVES[cv.name]
    K = pc.K
    q_inf = Q[2] == (0, 0) or Q[2] == 0
    p_inf = P[2] == 0
    if q_inf or p_inf:
        return tuple(K.one)
    if Q[2] != (1, 0) or P[2] != 1:
        raise ValueError("pairing expects normalised points")
    f = pc.miller(pc.untwist((Q[0], Q[1])), pc.cast_g1(P))
    return tuple(pc.final_exponentiate(f))
