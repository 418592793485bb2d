"""
oracle/plonk_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement, on plain Python integers, of the DETERMINISTIC pieces of the reference's
PLONK prover round (swusjask/kzg-snark plonk/prover.py and the plonk/encoder.py helpers it
calls), i.e. the harness of BASELINE config 5.  Everything the
reference samples at random (the eleven blinders b1..b11, plonk/prover.py:72-75 and :346) and
everything it derives from the transcript (beta, gamma, alpha, zeta) is an ARGUMENT here, so a
device prover's intermediate polynomials can be compared coefficient by coefficient.

Only tests/ may import this module (same rule as oracle/py_oracle.py).

PARITY UNPINNED, as for the rest of oracle/: the reference holds no value-level vectors for
these functions (its self-tests are randomised relation checks) and Sage is
absent.  What pins this file: the identities the reference itself asserts at
plonk/prover.py:110 (L1 * (z - 1) divisible by v_H), :171 (r(zeta) = 0), :354 (t equals the sum
of its parts) -- all re-checked in tests/test_plonk_oracle.py on the reference's own 16-gate
instance -- and the product of polynomials cross-checked against the schoolbook definition.

Polynomials are coefficient lists, low degree first, entries in [0, r).  Sage's dense
polynomial product is restated as an exact integer product (Kronecker substitution: the
coefficients are packed into one big integer, multiplied, unpacked, reduced mod r); that is a
way of evaluating the same definition c_k = sum_{i+j=k} a_i b_j, not a different algorithm.
"""
from . import py_oracle as O


# ---------------------------------------------------------------------------------------
# the slice of Sage's PolynomialRing the prover uses (plonk/prover.py:83-85, 240, 297-316)
# ---------------------------------------------------------------------------------------

def p_norm(c):
    return O.poly_normalize(c)


def p_add(a, b, r):
    n = max(len(a), len(b))
    return p_norm([((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % r for i in range(n)])


def p_sub(a, b, r):
    n = max(len(a), len(b))
    return p_norm([((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % r for i in range(n)])


def p_scale(a, s, r):
    s %= r
    return p_norm([x * s % r for x in a])


def p_mul_schoolbook(a, b, r):
    """Definition c_k = sum_{i+j=k} a_i b_j; used to check p_mul."""
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % r
    return p_norm(out)


def p_mul(a, b, r):
    """The same product through one big-integer multiplication."""
    a, b = p_norm(a), p_norm(b)
    if not a or not b:
        return []
    slot = 2 * r.bit_length() + min(len(a), len(b)).bit_length() + 1      # a column sum fits one slot
    nbytes = (slot + 7) // 8
    slot = nbytes * 8

    def pack(c):
        return int.from_bytes(b"".join(int(x).to_bytes(nbytes, "little") for x in c), "little")

    prod = pack(a) * pack(b)
    m = len(a) + len(b) - 1
    raw = prod.to_bytes(nbytes * (m + 1), "little")
    return p_norm([int.from_bytes(raw[k * nbytes:(k + 1) * nbytes], "little") % r for k in range(m)])


def p_eval(c, z, r):
    return O.poly_eval(c, z % r, r)


def p_shift_argument(c, g, r):
    """p(g X): coefficient i scaled by g^i (plonk/prover.py:305, z_poly(g * X))."""
    out, gp = [], 1
    for x in c:
        out.append(x * gp % r)
        gp = gp * g % r
    return p_norm(out)


def vanishing(n, r):
    """v_H = X^n - 1 (plonk/encoder.py:69)."""
    return [r - 1] + [0] * (n - 1) + [1]


def p_divmod_vanishing(c, n, r):
    """divmod(c, X^n - 1): q_i = c_(i+n) + q_(i+n), top down."""
    c = list(c)
    if len(c) <= n:
        return [], p_norm(c)
    q = [0] * (len(c) - n)
    for i in range(len(c) - n - 1, -1, -1):
        hi = q[i + n] if i + n < len(q) else 0
        q[i] = (c[i + n] + hi) % r
    rem = [(c[i] + (q[i] if i < len(q) else 0)) % r for i in range(n)]
    return p_norm(q), p_norm(rem)


def p_div_linear_exact(c, root, r):
    """c // (X - root) for a polynomial that vanishes at root (Sage `//`)."""
    q, rem = O.poly_divide_linear(c, root % r, r)
    assert rem == 0, "not divisible by (X - root)"
    return q


# ---------------------------------------------------------------------------------------
# plonk/encoder.py helpers the prover calls
# ---------------------------------------------------------------------------------------

def lagrange_basis(i, n, g, r):
    """plonk/encoder.py:215-235: L_i(X) = g^i (X^n - 1) // (n (X - g^i))."""
    gi = pow(g, i, r)
    numerator = p_scale(vanishing(n, r), gi, r)                 # :230
    q = p_div_linear_exact(numerator, gi, r)                    # :231-232, the (X - g^i) factor
    return p_scale(q, pow(n % r, -1, r), r)                     # ... and the constant n


def public_input_poly(x, n, g, r):
    """plonk/encoder.py:237-257: PI(X) = -sum_i x_i L_i(X)."""
    PI = []
    for i, x_i in enumerate(x):                                  # :252-255
        PI = p_sub(PI, p_scale(lagrange_basis(i, n, g, r), int(x_i), r), r)
    return PI


def first_lagrange(n, r):
    """plonk/prover.py:109 and :312: L1 = (X^n - 1) / (n (X - 1))."""
    return p_scale(p_div_linear_exact(vanishing(n, r), 1, r), pow(n % r, -1, r), r)


# ---------------------------------------------------------------------------------------
# plonk/prover.py restated
# ---------------------------------------------------------------------------------------

def wire_polynomial(values, g, n, b_x, b_0, r):
    """plonk/prover.py:83-85: (b1 X + b2) v_H + fft_ff_interpolation(values, g, Fq).
    b_x multiplies X (b1 / b3 / b5), b_0 is the constant (b2 / b4 / b6)."""
    interp = O.fft_ff_interpolation([int(v) % r for v in values], g, r)
    return p_add(p_mul([b_0 % r, b_x % r], vanishing(n, r), r), interp, r)


def permutation_values(a_values, b_values, c_values, sigma_star, beta, gamma, k1, k2, n, H, r):
    """plonk/prover.py:243-261: the accumulator z(w^i), n - 1 sequential field divisions."""
    z_values = [1]                                               # :243
    for i in range(n - 1):                                       # :245
        num = ((a_values[i] + beta * H[i] + gamma)
               * (b_values[i] + beta * k1 * H[i] + gamma)
               * (c_values[i] + beta * k2 * H[i] + gamma)) % r   # :247-249
        den = ((a_values[i] + beta * sigma_star[i] + gamma)
               * (b_values[i] + beta * sigma_star[i + n] + gamma)
               * (c_values[i] + beta * sigma_star[i + 2 * n] + gamma)) % r   # :252-254
        if den == 0:                                             # :256-258
            raise ValueError("Denominator is zero in permutation polynomial calculation")
        z_values.append(z_values[-1] * num % r * pow(den, -1, r) % r)       # :261
    return z_values


def permutation_polynomial(a_values, b_values, c_values, sigma_star, beta, gamma, g, k1, k2, n, H,
                           b7, b8, b9, r):
    """plonk/prover.py:214-269: z(X) = (b7 X^2 + b8 X + b9) v_H + interpolation of the accumulator."""
    blind = p_mul([b9 % r, b8 % r, b7 % r], vanishing(n, r), r)                 # :240
    z_values = permutation_values(a_values, b_values, c_values, sigma_star, beta, gamma, k1, k2, n, H, r)
    z_interp = O.fft_ff_interpolation(z_values, g, r)                            # :264
    return p_add(blind, z_interp, r)                                             # :267


def quotient_numerator(a, b, c, z, qM, qL, qR, qO, qC, S1, S2, S3, alpha, beta, gamma, PI, n, g, k1, k2, r):
    """The four numerators of plonk/prover.py:297-313, summed before the division by v_H."""
    X = [0, 1]

    def lin(p, s_poly, s):
        """p + s * s_poly + gamma."""
        return p_add(p_add(p, p_scale(s_poly, s, r), r), [gamma % r], r)

    term1 = p_add(p_add(p_add(p_mul(p_mul(a, b, r), qM, r), p_mul(a, qL, r), r),
                        p_add(p_mul(b, qR, r), p_mul(c, qO, r), r), r), p_add(PI, qC, r), r)      # :297
    term2 = p_scale(p_mul(p_mul(p_mul(z, lin(a, X, beta), r), lin(b, X, beta * k1), r),
                          lin(c, X, beta * k2), r), alpha, r)                                     # :300-302
    z_shifted = p_shift_argument(z, g, r)                                                          # :305
    term3 = p_scale(p_mul(p_mul(p_mul(lin(a, S1, beta), lin(b, S2, beta), r), lin(c, S3, beta), r),
                          z_shifted, r), -alpha, r)                                                # :306-309
    L1 = first_lagrange(n, r)                                                                      # :312
    term4 = p_scale(p_mul(p_sub(z, [1], r), L1, r), alpha * alpha, r)                              # :313
    return p_add(p_add(term1, term2, r), p_add(term3, term4, r), r)                                # :316


def quotient_polynomial(a, b, c, z, qM, qL, qR, qO, qC, S1, S2, S3, alpha, beta, gamma, PI, n, g, k1, k2, r):
    """plonk/prover.py:271-318.  The reference divides every term by v_H on its own and lets
    `R(...)` coerce the sum back into the polynomial ring (:316), which succeeds exactly when
    the summed numerator is divisible by v_H; the restatement divides the sum once and raises
    where the reference's coercion would."""
    t, rem = p_divmod_vanishing(
        quotient_numerator(a, b, c, z, qM, qL, qR, qO, qC, S1, S2, S3, alpha, beta, gamma, PI, n, g, k1, k2, r), n, r)
    if rem:
        raise ArithmeticError("quotient numerator is not divisible by v_H")
    return t


def split_quotient(t, n, b10, b11, r):
    """plonk/prover.py:320-356: t_lo + b10 X^n, t_mid - b10 + b11 X^n, t_hi - b11."""
    tc = list(t) + [0] * (3 * n - len(t))                        # :337-338
    t_lo = p_add(tc[:n], [0] * n + [b10 % r], r)                 # :349
    t_mid = p_add(p_sub(tc[n:2 * n], [b10 % r], r), [0] * n + [b11 % r], r)      # :350
    t_hi = p_sub(tc[2 * n:], [b11 % r], r)                       # :351
    recombined = p_add(p_add(t_lo, [0] * n + t_mid, r), [0] * (2 * n) + t_hi, r)
    assert recombined == p_norm(t), "t(X) does not equal the sum of its parts"   # :354
    return t_lo, t_mid, t_hi


def linearization_polynomial(a_zeta, b_zeta, c_zeta, s1_zeta, s2_zeta, z_omega_zeta,
                             qM, qL, qR, qO, qC, S3, z, t_lo, t_mid, t_hi,
                             alpha, beta, gamma, zeta, PI, n, k1, k2, r):
    """plonk/prover.py:358-414."""
    z_H_zeta = (pow(zeta, n, r) - 1) % r                                         # :383
    L1_zeta = z_H_zeta * pow(n * (zeta - 1) % r, -1, r) % r                      # :386
    PI_zeta = p_eval(PI, zeta, r)                                                # :389
    term1 = p_add(p_add(p_add(p_scale(qM, a_zeta * b_zeta, r), p_scale(qL, a_zeta, r), r),
                        p_add(p_scale(qR, b_zeta, r), p_scale(qO, c_zeta, r), r), r),
                  p_add([PI_zeta], qC, r), r)                                    # :393
    f1 = ((a_zeta + beta * zeta + gamma) * (b_zeta + beta * k1 * zeta + gamma)
          * (c_zeta + beta * k2 * zeta + gamma)) % r
    term2 = p_scale(z, alpha * f1, r)                                            # :396-398
    f2 = (a_zeta + beta * s1_zeta + gamma) * (b_zeta + beta * s2_zeta + gamma) % r
    inner = p_add(p_scale(S3, beta, r), [(c_zeta + gamma) % r], r)
    term3 = p_scale(inner, -alpha * f2 * z_omega_zeta, r)                        # :401-403
    term4 = p_scale(p_sub(z, [1], r), alpha * alpha * L1_zeta, r)                # :406
    zn = pow(zeta, n, r)
    t_sum = p_add(p_add(t_lo, p_scale(t_mid, zn, r), r), p_scale(t_hi, zn * zn, r), r)
    return p_sub(p_add(p_add(term1, term2, r), p_add(term3, term4, r), r),
                 p_scale(t_sum, z_H_zeta, r), r)                                 # :410-412


# ---------------------------------------------------------------------------------------
# transcript.py restated, and the whole round (plonk/prover.py:24-212) with tau and the
# blinders supplied: a deterministic proof that tests/golden/ freezes
# ---------------------------------------------------------------------------------------
import hashlib
import struct


class FieldElement(int):
    """A field element as the transcript sees it: str() is the decimal value, which is what
    `str(data).encode()` makes of a Sage IntegerMod (transcript.py:80-85).  A plain Python int
    is serialised differently (struct ">q", transcript.py:70-71), hence the distinct type."""
    __slots__ = ()

    def __str__(self):
        return int.__repr__(self)

    __repr__ = __str__


class Transcript:
    """transcript.py:4-100."""

    def __init__(self, label, r):
        self.r = r
        self.state = hashlib.sha256(label.encode()).digest()                     # :23

    def _serialize(self, data):                                                  # :58-85
        if isinstance(data, FieldElement):
            return str(data).encode()
        if isinstance(data, str):
            return data.encode()
        if isinstance(data, int):
            return struct.pack(">q", data)
        if isinstance(data, bytes):
            return data
        if isinstance(data, list):
            return b"".join(self._serialize(item) for item in data)
        return str(data).encode()

    def _update_state(self, label, data):                                        # :87-100
        self.state = hashlib.sha256(self.state + label.encode() + data).digest()

    def append_message(self, label, data):                                       # :25-34
        self._update_state(label, self._serialize(data))

    def get_challenge(self, label):                                              # :36-56
        challenge_state = hashlib.sha256(self.state + label.encode()).digest()
        challenge = int.from_bytes(challenge_state, byteorder="big") % self.r
        self._update_state(label, challenge_state)
        return challenge


def prove_round(circuit, n, g, k1, k2, tau, blinders, cv):
    """plonk/prover.py:24-212 end to end on the oracle: interpolation by O.fft_ff_interpolation,
    commitments and openings by O.commit / O.open_ against O.setup(n + 5, tau) (main.py:85).
    Points enter the transcript as the normalised tuple (x, y, 1) -- the representative this
    engine's facade returns (the reference hashes py_ecc's un-normalised
    projective triple, which no other group-law implementation can reproduce; prover and
    verifier only need to agree with each other).  Public inputs are plain ints, evaluations
    are field elements, exactly the types plonk/prover.py:57 and :155 hand to the transcript.
    Returns (proof dict of ints / (x, y, 1) tuples, challenges, polynomials)."""
    r = cv.r
    qM, qL, qR, qO, qC, perm, x, w = circuit
    assert len(qM) == n and len(perm) == 3 * n
    full = [int(v) % r for v in list(x) + list(w)]
    cols = [full[i * n:(i + 1) * n] for i in range(3)]
    H = [pow(g, i, r) for i in range(n)]
    label = H + [k1 * h % r for h in H] + [k2 * h % r for h in H]                # plonk/encoder.py:139-147
    sigma_star = [label[perm[i]] for i in range(3 * n)]
    interp = lambda vals: O.fft_ff_interpolation([int(v) % r for v in vals], g, r)   # noqa: E731
    sel = {k: interp(v) for k, v in (("qM", qM), ("qL", qL), ("qR", qR), ("qO", qO), ("qC", qC))}
    S = [interp(sigma_star[i * n:(i + 1) * n]) for i in range(3)]
    ck = O.setup(n + 5, tau, cv)

    def point(p):
        a = O.normalize(p, cv)
        return (1, 1, 0) if a is None else (a[0], a[1], 1)

    commit = lambda polys: [point(c) for c in O.commit(ck, polys, cv)]            # noqa: E731
    b = [int(v) % r for v in blinders]
    tr = Transcript("plonk-proof", r)                                            # :54
    tr.append_message("public-inputs", list(x))                                  # :57
    PI = public_input_poly(x, n, g, r)                                           # :68
    a_p = wire_polynomial(cols[0], g, n, b[0], b[1], r)                          # :83-85
    b_p = wire_polynomial(cols[1], g, n, b[2], b[3], r)
    c_p = wire_polynomial(cols[2], g, n, b[4], b[5], r)
    wire_comms = commit([a_p, b_p, c_p])                                         # :89
    tr.append_message("round1-commitments", wire_comms)                          # :93
    beta, gamma = tr.get_challenge("beta"), tr.get_challenge("gamma")            # :97-98
    z_p = permutation_polynomial(cols[0], cols[1], cols[2], sigma_star, beta, gamma, g, k1, k2, n, H,
                                 b[6], b[7], b[8], r)                            # :101-106
    _, rem = p_divmod_vanishing(p_mul(first_lagrange(n, r), p_sub(z_p, [1], r), r), n, r)
    assert rem == [], "z_poly does not satisfy L1 condition"                     # :109-110
    z_comm = commit([z_p])[0]                                                    # :113
    tr.append_message("round2-commitment", z_comm)                               # :116
    alpha = tr.get_challenge("alpha")                                            # :120
    t_p = quotient_polynomial(a_p, b_p, c_p, z_p, sel["qM"], sel["qL"], sel["qR"], sel["qO"], sel["qC"],
                              S[0], S[1], S[2], alpha, beta, gamma, PI, n, g, k1, k2, r)   # :123-129
    t_lo, t_mid, t_hi = split_quotient(t_p, n, b[9], b[10], r)                   # :132
    t_comms = commit([t_lo, t_mid, t_hi])                                        # :136
    tr.append_message("round3-commitments", t_comms)                             # :140
    zeta = tr.get_challenge("zeta")                                              # :144
    ev = {"a": p_eval(a_p, zeta, r), "b": p_eval(b_p, zeta, r), "c": p_eval(c_p, zeta, r),
          "s_sigma1": p_eval(S[0], zeta, r), "s_sigma2": p_eval(S[1], zeta, r),
          "z_omega": p_eval(z_p, zeta * g % r, r)}                               # :147-152
    order = ("a", "b", "c", "s_sigma1", "s_sigma2", "z_omega")
    tr.append_message("round4-evaluations", [FieldElement(ev[k]) for k in order])   # :155-156
    v = tr.get_challenge("v")                                                    # :160
    r_p = linearization_polynomial(ev["a"], ev["b"], ev["c"], ev["s_sigma1"], ev["s_sigma2"], ev["z_omega"],
                                   sel["qM"], sel["qL"], sel["qR"], sel["qO"], sel["qC"], S[2], z_p,
                                   t_lo, t_mid, t_hi, alpha, beta, gamma, zeta, PI, n, k1, k2, r)   # :163-168
    assert p_eval(r_p, zeta, r) == 0, "r(zeta) should be zero"                   # :171
    W_z = point(O.open_(ck, [r_p, a_p, b_p, c_p, S[0], S[1]], zeta, v, cv)[0])   # :174-184
    W_zw = point(O.open_(ck, [z_p], zeta * g % r, v, cv)[0])                     # :185
    proof = {"commitments": dict(zip(("a", "b", "c"), wire_comms), z=z_comm,
                                 t_lo=t_comms[0], t_mid=t_comms[1], t_hi=t_comms[2]),
             "evaluations": ev, "kzg_proofs": {"W_z": W_z, "W_zw": W_zw}}        # :188-210
    challenges = {"beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v}
    polys = {"a": a_p, "b": b_p, "c": c_p, "z": z_p, "PI": PI, "t": t_p, "t_lo": t_lo, "t_mid": t_mid,
             "t_hi": t_hi, "r": r_p}
    return proof, challenges, polys
