"""
oracle/py_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement, on plain Python integers, of the reference's hot path
(swusjask/kzg-snark: fft_ff.py and kzg.py commit/open/setup).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the shipped package (kzg_snark_amd/) never does.

PARITY UNPINNED: the reference holds no golden vectors or known-answer tests
for this path (every self-test samples unseeded
randomness and checks a relation), and its arithmetic lives in two un-vendored,
un-pinned third-party packages that are absent here:
  * SageMath (GF(r) elements, dense polynomials)     -- kzg.py:1, fft_ff.py:1
  * py_ecc   (optimized_bn128 / optimized_bls12_381) -- kzg.py:26-35
What pins this oracle instead (tests/test_oracle_pins.py):
  * the curve constants published in the curve specifications (EIP-196/197 for
    BN254 a.k.a. alt_bn128; the BLS12-381 spec / IETF pairing-friendly-curves
    draft) and the BN254 scalar modulus embedded in both of the reference's
    constraint-system/*.pkl fixtures;
  * public known-answer points (2*G1 on both curves);
  * algebraic identities (r*G = O, trapdoor identity commit(p) = p(tau)*G,
    NTT vs the O(n^2) DFT definition, INTT(NTT(x)) = x).

The G1 group law restates the *published algorithm* of py_ecc's
optimized_curve modules (homogeneous projective (x, y, z), infinity = (1,1,0),
recursive double-and-add `multiply`); it is written from the textbook formulas
those modules implement, recalled rather than read (py_ecc is not present),
so un-normalised projective triples are NOT claimed to match py_ecc's; affine
points are representation independent and are what parity is defined on.
"""

# --------------------------------------------------------------------------
# Curve parameters (public constants; see module docstring for provenance)
# --------------------------------------------------------------------------

class CurveParams:
    def __init__(self, name, p, r, b, g1, two_adicity, root_of_unity_gen):
        self.name = name
        self.p = p              # base field modulus (G1 coordinates)
        self.r = r              # scalar field modulus == curve_order (kzg.py:49)
        self.b = b              # y^2 = x^3 + b
        self.g1 = g1            # affine generator
        self.two_adicity = two_adicity
        self.mult_gen = root_of_unity_gen  # smallest primitive root of Fr

    def root_of_unity(self, n):
        """What Sage's Fq(1).nth_root(n) is recalled to return for callers such as
        plonk/encoder.py:49 -- g^((r-1)/n) with g the smallest primitive root."""
        assert n & (n - 1) == 0 and n <= (1 << self.two_adicity)
        return pow(self.mult_gen, (self.r - 1) // n, self.r)


BN254 = CurveParams(
    "bn254",
    p=21888242871839275222246405745257275088696311157297823662689037894645226208583,
    r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
    b=3,
    g1=(1, 2),
    two_adicity=28,
    root_of_unity_gen=5,
)

BLS12_381 = CurveParams(
    "bls12_381",
    p=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    r=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    b=4,
    g1=(
        0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
        0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,
    ),
    two_adicity=32,
    root_of_unity_gen=7,
)

CURVES = {"bn254": BN254, "bls12_381": BLS12_381}


def curve(curve_type):
    """Curve selection with the reference's error behaviour (kzg.py:26-37)."""
    if curve_type not in CURVES:
        raise ValueError(f"Unsupported curve type: {curve_type}")
    return CURVES[curve_type]


# --------------------------------------------------------------------------
# fft_ff.py restated on ints mod r
# --------------------------------------------------------------------------

def fft_ff(coeffs, w, r):
    """fft_ff.py:3-37.  Recursive radix-2 DIT; natural order in and out.
    n == 1 returns the input list object itself (fft_ff.py:16-17).  No length
    or primitivity check, exactly like the reference."""
    n = len(coeffs)
    if n == 1:
        return coeffs
    even = coeffs[0::2]                       # fft_ff.py:20
    odd = coeffs[1::2]                        # fft_ff.py:21
    w_squared = w * w % r                     # fft_ff.py:24
    even_fft = fft_ff(even, w_squared, r)     # fft_ff.py:25
    odd_fft = fft_ff(odd, w_squared, r)       # fft_ff.py:26
    result = [0] * n                          # fft_ff.py:29
    w_power = 1                               # fft_ff.py:30
    for i in range(n // 2):                   # fft_ff.py:32-35
        t = w_power * odd_fft[i] % r
        result[i] = (even_fft[i] + t) % r
        result[i + n // 2] = (even_fft[i] - t) % r
        w_power = w_power * w % r
    return result


def ifft_ff(values, w, r):
    """fft_ff.py:39-58: forward transform with w^-1, then scale by n^-1."""
    n = len(values)
    w_inv = pow(w, -1, r)                     # fft_ff.py:53
    result = fft_ff(values, w_inv, r)         # fft_ff.py:54
    n_inv = pow(n % r, -1, r)                 # fft_ff.py:57
    return [x * n_inv % r for x in result]    # fft_ff.py:58


def multiplicative_order_at_least(g, n, r):
    """Stand-in for `g.multiplicative_order() >= n` (fft_ff.py:77-78) for the
    power-of-two n the assert at :74 has already established: the order of g is
    >= n unless g^(m) == 1 for some m < n; for the 2-power-order elements the
    callers pass it suffices to test g^(n/2) != 1.  For general g we fall back
    to the definition (small n only)."""
    if n == 1:
        return True
    g %= r
    if g == 0:
        raise ArithmeticError("multiplicative order of 0 is undefined")
    if pow(g, n // 2, r) != 1 and pow(g, n, r) == 1:
        return True      # order divides n but not n/2 => order == n
    # general case: order >= n  <=>  g^k != 1 for 1 <= k < n
    acc = 1
    for _ in range(1, n):
        acc = acc * g % r
        if acc == 1:
            return False
    return True


def fft_ff_interpolation(values, g, r):
    """fft_ff.py:60-85.  Returns the coefficient list of the Sage polynomial the
    reference builds (R(coeffs) drops trailing zeros; the zero polynomial is [])."""
    n = len(values)
    assert (n & (n - 1)) == 0, "Length of values must be a power of 2"   # :74
    assert multiplicative_order_at_least(g, n, r), \
        f"Order of g must be at least n ({n})"                           # :77-78
    coeffs = ifft_ff(values, g, r)                                       # :81
    return poly_normalize(coeffs)                                        # :84-85


def dft_naive(coeffs, w, r):
    """Definition out[k] = sum_j c[j] w^(jk): independent check of fft_ff for
    primitive w."""
    n = len(coeffs)
    return [sum(c * pow(w, j * k, r) for j, c in enumerate(coeffs)) % r for k in range(n)]


# --------------------------------------------------------------------------
# Dense polynomials as coefficient lists (the slice of Sage's PolynomialRing
# that kzg.py:93-110,137-154 uses)
# --------------------------------------------------------------------------

def poly_normalize(c):
    c = list(c)
    while c and c[-1] == 0:
        c.pop()
    return c


def poly_degree(c):
    """Sage: degree of the zero polynomial is -1."""
    return len(poly_normalize(c)) - 1


def poly_eval(c, z, r):
    acc = 0
    for a in reversed(c):
        acc = (acc * z + a) % r
    return acc


def poly_divide_linear(c, z, r):
    """(p(X) - p(z)) // (X - z) by synthetic division (kzg.py:154); returns
    (quotient coefficients, p(z))."""
    c = poly_normalize(c)
    if not c:
        return [], 0
    q = [0] * (len(c) - 1)
    carry = 0
    for i in range(len(c) - 1, 0, -1):
        carry = (c[i] + carry * z) % r
        q[i - 1] = carry
    pz = (c[0] + carry * z) % r
    return poly_normalize(q), pz


# --------------------------------------------------------------------------
# G1 group law, py_ecc-shaped (homogeneous projective, Python ints mod p)
# --------------------------------------------------------------------------

def Z1():
    return (1, 1, 0)


def is_inf(pt):
    return pt[2] == 0


def from_affine(xy):
    return (xy[0], xy[1], 1)


def double(pt, cv):
    p = cv.p
    x, y, z = pt
    W = 3 * x * x % p
    S = y * z % p
    B = x * y % p * S % p
    H = (W * W - 8 * B) % p
    S_squared = S * S % p
    newx = 2 * H * S % p
    newy = (W * (4 * B - H) - 8 * y * y % p * S_squared) % p
    newz = 8 * S * S_squared % p
    return (newx, newy, newz)


def add(p1, p2, cv):
    p = cv.p
    if p1[2] == 0 or p2[2] == 0:
        return p1 if p2[2] == 0 else p2
    x1, y1, z1 = p1
    x2, y2, z2 = p2
    U1 = y2 * z1 % p
    U2 = y1 * z2 % p
    V1 = x2 * z1 % p
    V2 = x1 * z2 % p
    if V1 == V2 and U1 == U2:
        return double(p1, cv)
    elif V1 == V2:
        return (1, 1, 0)
    U = (U1 - U2) % p
    V = (V1 - V2) % p
    V_squared = V * V % p
    V_squared_times_V2 = V_squared * V2 % p
    V_cubed = V * V_squared % p
    W = z1 * z2 % p
    A = (U * U % p * W - V_cubed - 2 * V_squared_times_V2) % p
    newx = V * A % p
    newy = (U * (V_squared_times_V2 - A) - V_cubed * U2) % p
    newz = V_cubed * W % p
    return (newx, newy, newz)


def neg(pt, cv):
    x, y, z = pt
    return (x, (-y) % cv.p, z)


def multiply(pt, n, cv):
    """py_ecc's published double-and-add: multiply(pt, n) = pt if n == 1;
    multiply(double(pt), n/2) if n even; add(multiply(double(pt), n//2), pt) if
    odd.  Unrolled: the deepest call returns 2^k*pt (k = top bit) and the adds
    of the lower set bits are applied on the way back out, highest first."""
    n = int(n)
    if n == 0:
        return (1, 1, 0)
    doubles = [pt]
    for _ in range(n.bit_length() - 1):
        doubles.append(double(doubles[-1], cv))
    acc = doubles[-1]
    for i in range(n.bit_length() - 2, -1, -1):
        if (n >> i) & 1:
            acc = add(acc, doubles[i], cv)
    return acc


def normalize(pt, cv):
    """Affine (x, y) or None for the point at infinity."""
    if pt[2] == 0:
        return None
    zinv = pow(pt[2], -1, cv.p)
    return (pt[0] * zinv % cv.p, pt[1] * zinv % cv.p)


def eq(p1, p2, cv):
    x1, y1, z1 = p1
    x2, y2, z2 = p2
    if z1 == 0 or z2 == 0:
        return z1 == 0 and z2 == 0
    return (x1 * z2 - x2 * z1) % cv.p == 0 and (y1 * z2 - y2 * z1) % cv.p == 0


def is_on_curve(xy, cv):
    if xy is None:
        return True
    x, y = xy
    return (y * y - x * x * x - cv.b) % cv.p == 0


# --------------------------------------------------------------------------
# kzg.py restated
# --------------------------------------------------------------------------

def setup(max_degree, tau, cv):
    """kzg.py:56-78 with tau supplied by the caller (the reference samples it
    unseeded at :67 and discards it).  Returns ([tau^i G1], tau) -- the G2 half of
    the key is outside the hot path."""
    g = from_affine(cv.g1)
    ck = [g]
    t = 1
    for _ in range(1, max_degree + 1):
        t = t * tau % cv.r
        ck.append(multiply(g, t, cv))          # kzg.py:71-72
    return ck


def commit(ck, polynomials, cv):
    """kzg.py:80-120: per polynomial, sum_i p_i * ck[i] by independent
    scalar-muls and a running add; zero coefficients skipped (:113-114);
    ValueError when degree > len(ck)-1 (:103-106)."""
    max_degree = len(ck) - 1
    out = []
    for poly in polynomials:
        coeffs = poly_normalize([int(c) % cv.r for c in poly])
        deg = len(coeffs) - 1
        if deg > max_degree:
            raise ValueError(
                f"Polynomial degree {deg} exceeds maximum allowed degree {max_degree}")
        acc = Z1()                                           # kzg.py:109
        for i, c in enumerate(coeffs):                       # kzg.py:112-116
            if c == 0:
                continue
            acc = add(acc, multiply(ck[i], c, cv), cv)
        out.append(acc)
    return out


def combine(polynomials, xi, r):
    """kzg.py:147-150: sum_i xi^(i+1) * p_i  (first polynomial scaled by xi, not 1)."""
    n = max((len(p) for p in polynomials), default=0)
    acc = [0] * n
    xp = 1
    for poly in polynomials:
        xp = xp * xi % r
        for j, c in enumerate(poly):
            acc[j] = (acc[j] + xp * (int(c) % r)) % r
    return poly_normalize(acc)


def open_(ck, polynomials, z, xi, cv):
    """kzg.py:122-159.  Returns (proof point, combined evaluation P(z))."""
    z = int(z) % cv.r
    xi = int(xi) % cv.r
    combined = combine(polynomials, xi, cv.r)
    witness, pz = poly_divide_linear(combined, z, cv.r)
    return commit(ck, [witness], cv)[0], pz


def commit_trapdoor(poly, tau, cv):
    """commit(ck, p) == p(tau) * G1 when ck = setup(tau)."""
    return multiply(from_affine(cv.g1), poly_eval([int(c) % cv.r for c in poly], tau, cv.r), cv)


def open_trapdoor(polynomials, z, xi, tau, cv):
    """open == ((P(tau) - P(z)) / (tau - z)) * G1."""
    r = cv.r
    comb = combine(polynomials, int(xi) % r, r)
    num = (poly_eval(comb, tau, r) - poly_eval(comb, int(z) % r, r)) % r
    s = num * pow((tau - z) % r, -1, r) % r
    return multiply(from_affine(cv.g1), s, cv)
