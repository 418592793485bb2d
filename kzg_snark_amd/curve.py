"""Single-point G1/G2 helpers with the py_ecc call shapes the reference binds at
kzg.py:27-49 (G1, G2, Z1, Z2, multiply, add, neg, eq, curve_order).

These serve the O(#polynomials) host-side group operations of the verifiers
(plonk/verifier.py:134-178, marlin/verifier.py:107-141) and the tau*G2 half of
setup; they are NOT on the data-parallel path -- commit/open/SRS generation run
on the GPU (csrc/msm.hip) and nothing here is a fallback for them.

Points are 3-tuples like py_ecc's: (x, y, z) with z = 1 for finite points (always
normalised, so equal points have equal tuples and equal transcript bytes) and
(1, 1, 0) for infinity.  G1 coordinates are Python ints; G2 coordinates are
Fp2 pairs (c0, c1) = c0 + c1*u with u^2 = -1."""


class CurveDef:
    def __init__(self, name, p, r, b, g1, g2, b2):
        self.name, self.p, self.r, self.b, self.g1, self.g2, self.b2 = name, p, r, b, g1, g2, b2


_BN_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
_BN_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
_BLS_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def _fp2_mul(a, b, p):
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def _fp2_inv(a, p):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, p)
    return (a[0] * d % p, (-a[1]) * d % p)


BN254 = CurveDef(
    "bn254", _BN_P, _BN_R, 3, (1, 2),
    ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
      11559732032986387107991004021392285783925812861821192530917403151452391805634),
     (8495653923123431417604973247489272438418190587263600148770280649306958101930,
      4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    _fp2_mul((3, 0), _fp2_inv((9, 1), _BN_P), _BN_P),        # 3 / (9 + u)
)

BLS12_381 = CurveDef(
    "bls12_381", _BLS_P, _BLS_R, 4,
    (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
     0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
    ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
      0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
     (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
      0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)),
    (4, 4),                                                  # 4 * (1 + u)
)

CURVES = {"bn254": BN254, "bls12_381": BLS12_381}


class _Fp:
    """Arithmetic namespace for the base field (ints)."""
    def __init__(self, p): self.p = p
    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def mul(self, a, b): return a * b % self.p
    def neg(self, a): return (-a) % self.p
    def inv(self, a): return pow(a, -1, self.p)
    def is_zero(self, a): return a % self.p == 0
    zero = 0
    one = 1
    def small(self, k): return k % self.p


class _Fp2:
    """Arithmetic namespace for Fp2 = Fp[u]/(u^2+1) (pairs)."""
    def __init__(self, p): self.p = p
    def add(self, a, b): return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)
    def sub(self, a, b): return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)
    def mul(self, a, b): return _fp2_mul(a, b, self.p)
    def neg(self, a): return ((-a[0]) % self.p, (-a[1]) % self.p)
    def inv(self, a): return _fp2_inv(a, self.p)
    def is_zero(self, a): return a[0] % self.p == 0 and a[1] % self.p == 0
    zero = (0, 0)
    one = (1, 0)
    def small(self, k): return (k % self.p, 0)


class Group:
    """Short-Weierstrass group y^2 = x^3 + b over a field namespace K (affine
    arithmetic with one inversion per operation: these are single-point helpers)."""

    def __init__(self, K, order):
        self.K = K
        self.order = order
        self.Z = (K.one, K.one, K.zero)      # py_ecc's Z1 / Z2 shape

    def is_inf(self, pt):
        return self.K.is_zero(pt[2])

    def normalize(self, pt):
        K = self.K
        if self.is_inf(pt):
            return self.Z
        if pt[2] == K.one:
            return (pt[0], pt[1], K.one)
        zi = K.inv(pt[2])
        return (K.mul(pt[0], zi), K.mul(pt[1], zi), K.one)

    def neg(self, pt):
        if self.is_inf(pt):
            return self.Z
        return (pt[0], self.K.neg(pt[1]), pt[2])

    def double(self, pt):
        K = self.K
        pt = self.normalize(pt)
        if self.is_inf(pt) or K.is_zero(pt[1]):
            return self.Z
        x, y, _ = pt
        lam = K.mul(K.mul(K.small(3), K.mul(x, x)), K.inv(K.add(y, y)))
        x3 = K.sub(K.mul(lam, lam), K.add(x, x))
        y3 = K.sub(K.mul(lam, K.sub(x, x3)), y)
        return (x3, y3, K.one)

    def add(self, p1, p2):
        K = self.K
        if self.is_inf(p1):
            return self.normalize(p2)
        if self.is_inf(p2):
            return self.normalize(p1)
        p1, p2 = self.normalize(p1), self.normalize(p2)
        x1, y1, _ = p1
        x2, y2, _ = p2
        if x1 == x2:
            if y1 == y2:
                return self.double(p1)
            return self.Z
        lam = K.mul(K.sub(y2, y1), K.inv(K.sub(x2, x1)))
        x3 = K.sub(K.sub(K.mul(lam, lam), x1), x2)
        y3 = K.sub(K.mul(lam, K.sub(x1, x3)), y1)
        return (x3, y3, K.one)

    def multiply(self, pt, n):
        n = int(n)
        if n < 0:
            return self.multiply(self.neg(pt), -n)
        if n == 0 or self.is_inf(pt):
            return self.Z
        # Jacobian double-and-add, one inversion at the end
        K = self.K
        X, Y, _ = self.normalize(pt)
        ax, ay = X, Y
        rx = ry = rz = None
        for bit in bin(n)[2:]:
            if rx is not None:
                rx, ry, rz = self._jdbl(rx, ry, rz)
            if bit == "1":
                if rx is None:
                    rx, ry, rz = ax, ay, K.one
                else:
                    rx, ry, rz = self._jmadd(rx, ry, rz, ax, ay)
        if rx is None or K.is_zero(rz):
            return self.Z
        zi = K.inv(rz)
        zi2 = K.mul(zi, zi)
        return (K.mul(rx, zi2), K.mul(ry, K.mul(zi2, zi)), K.one)

    def _jdbl(self, X, Y, Z):
        K = self.K
        if K.is_zero(Z) or K.is_zero(Y):
            return K.one, K.one, K.zero
        A = K.mul(X, X); B = K.mul(Y, Y); Cc = K.mul(B, B)
        t = K.add(X, B)
        D = K.sub(K.sub(K.mul(t, t), A), Cc); D = K.add(D, D)
        E = K.add(K.add(A, A), A)
        F = K.mul(E, E)
        X3 = K.sub(F, K.add(D, D))
        C8 = K.add(Cc, Cc); C8 = K.add(C8, C8); C8 = K.add(C8, C8)
        Y3 = K.sub(K.mul(E, K.sub(D, X3)), C8)
        Z3 = K.mul(K.add(Y, Y), Z)
        return X3, Y3, Z3

    def _jmadd(self, X1, Y1, Z1, x2, y2):
        K = self.K
        if K.is_zero(Z1):
            return x2, y2, K.one
        Z1Z1 = K.mul(Z1, Z1)
        U2 = K.mul(x2, Z1Z1)
        S2 = K.mul(K.mul(y2, Z1), Z1Z1)
        H = K.sub(U2, X1)
        Rr = K.sub(S2, Y1)
        if K.is_zero(H):
            if K.is_zero(Rr):
                return self._jdbl(X1, Y1, Z1)
            return K.one, K.one, K.zero
        HH = K.mul(H, H)
        HHH = K.mul(H, HH)
        V = K.mul(X1, HH)
        X3 = K.sub(K.sub(K.mul(Rr, Rr), HHH), K.add(V, V))
        Y3 = K.sub(K.mul(Rr, K.sub(V, X3)), K.mul(Y1, HHH))
        Z3 = K.mul(Z1, H)
        return X3, Y3, Z3

    def eq(self, p1, p2):
        return self.normalize(p1) == self.normalize(p2)


def g1_group(cv):
    return Group(_Fp(cv.p), cv.r)


def g2_group(cv):
    return Group(_Fp2(cv.p), cv.r)


def on_curve_g1(pt, cv):
    if pt[2] == 0:
        return True
    x, y = pt[0], pt[1]
    return (y * y - x * x * x - cv.b) % cv.p == 0


def on_curve_g2(pt, cv):
    K = _Fp2(cv.p)
    if K.is_zero(pt[2]):
        return True
    x, y = pt[0], pt[1]
    return K.sub(K.mul(y, y), K.add(K.mul(K.mul(x, x), x), cv.b2)) == (0, 0)
