"""Sage-free stand-ins for the slice of SageMath the reference's hot path touches:
`GF(r)` elements (kzg.py:52, fft_ff.py `F`) and dense univariate polynomials
(`PolynomialRing(F, 'X')`, kzg.py:53-54, fft_ff.py:84-85).

Host-side glue only -- scalar arithmetic on single values (challenges, roots of
unity) and the polynomial container handed between encoder, prover and the
engine.  Bulk work (transforms, commitments, quotient-by-(X-z)) goes to the HIP
library; nothing here is a fallback for it."""
import random as _random


class FieldElement:
    __slots__ = ("v", "F")

    def __init__(self, v, F):
        self.v = v
        self.F = F

    def _c(self, o):
        if isinstance(o, FieldElement):
            return o.v
        return int(o) % self.F.p

    def __add__(self, o):
        if isinstance(o, Polynomial):
            return NotImplemented            # Polynomial.__radd__ takes over
        return FieldElement((self.v + self._c(o)) % self.F.p, self.F)
    __radd__ = __add__
    def __sub__(self, o):
        if isinstance(o, Polynomial):
            return NotImplemented            # Polynomial.__rsub__
        return FieldElement((self.v - self._c(o)) % self.F.p, self.F)
    def __rsub__(self, o): return FieldElement((self._c(o) - self.v) % self.F.p, self.F)
    def __mul__(self, o):
        if isinstance(o, Polynomial):
            return o.__rmul__(self)
        return FieldElement(self.v * self._c(o) % self.F.p, self.F)
    __rmul__ = __mul__
    def __neg__(self): return FieldElement((-self.v) % self.F.p, self.F)
    def __truediv__(self, o): return FieldElement(self.v * pow(self._c(o), -1, self.F.p) % self.F.p, self.F)
    def __rtruediv__(self, o): return FieldElement(self._c(o) * pow(self.v, -1, self.F.p) % self.F.p, self.F)
    def __pow__(self, e):
        e = int(e)
        return FieldElement(pow(self.v, e, self.F.p), self.F)     # negative e => modular inverse (w**(-1), fft_ff.py:53)
    def __eq__(self, o):
        if isinstance(o, FieldElement):
            return self.v == o.v and self.F.p == o.F.p
        try:
            return self.v == int(o) % self.F.p
        except (TypeError, ValueError):
            return NotImplemented
    def __hash__(self): return hash((self.v, self.F.p))
    def __int__(self): return self.v
    __index__ = __int__
    def __bool__(self): return self.v != 0
    def __repr__(self): return str(self.v)

    def multiplicative_order(self):
        """Order of the element (fft_ff.py:77).  Fast for the 2-power-order roots
        the encoders pass; general elements use the factorisation of p-1 that
        the field was given (or trial division for small cofactors)."""
        if self.v == 0:
            raise ArithmeticError("multiplicative order of 0 is not defined")
        p = self.F.p
        order = p - 1
        for q in self.F._order_factors():
            while order % q == 0 and pow(self.v, order // q, p) == 1:
                order //= q
        return order

    def is_zero(self): return self.v == 0


class PrimeField:
    """GF(p) for prime p (kzg.py:52 `GF(curve_order)`)."""

    def __init__(self, p):
        self.p = int(p)
        self._factors = None

    def __call__(self, x=0):
        if isinstance(x, FieldElement):
            return x if x.F.p == self.p else FieldElement(x.v % self.p, self)
        return FieldElement(int(x) % self.p, self)

    def elements(self, canonical):
        """Field elements for a list of ints already in [0, p) (what the engine returns): the bulk form of
        [F(v) for v in canonical] without the per-element coercion."""
        from itertools import repeat
        return list(map(FieldElement, canonical, repeat(self)))

    def order(self): return self.p
    def characteristic(self): return self.p
    cardinality = order
    def __eq__(self, o): return isinstance(o, PrimeField) and o.p == self.p
    def __hash__(self): return hash(("GF", self.p))
    def __repr__(self): return f"Finite Field of size {self.p}"

    def random_element(self):
        return FieldElement(_random.SystemRandom().randrange(self.p), self)

    def zero(self): return self(0)
    def one(self): return self(1)

    def _order_factors(self):
        """Prime factors of p-1 (trial division up to 2^20, remaining cofactor
        checked for primality by Miller-Rabin; enough for both curves' r-1)."""
        if self._factors is None:
            n = self.p - 1
            fs = []
            d = 2
            while d * d <= n and d < (1 << 20):
                if n % d == 0:
                    fs.append(d)
                    while n % d == 0:
                        n //= d
                d += 1 if d == 2 else 2
            if n > 1:
                fs.append(n)     # prime or a product of large primes; order is then an upper bound
            self._factors = fs
        return self._factors

    def multiplicative_generator(self):
        g = 2
        while True:
            if all(pow(g, (self.p - 1) // q, self.p) != 1 for q in self._order_factors()):
                return self(g)
            g += 1

    def root_of_unity(self, n):
        """Primitive n-th root as Sage's Fq(1).nth_root(n) is recalled to return it
        (plonk/encoder.py:49): g^((p-1)/n) for the smallest primitive root g."""
        n = int(n)
        assert (self.p - 1) % n == 0
        return self.multiplicative_generator() ** ((self.p - 1) // n)


def GF(p):
    return PrimeField(p)


class Polynomial:
    """Dense univariate polynomial over a PrimeField; coefficients low to high,
    no trailing zeros (Sage's normal form: R(coeffs) drops them, degree(0) = -1)."""
    __slots__ = ("c", "R")

    def __init__(self, coeffs, R):
        p = R.F.p
        c = [int(x) % p for x in coeffs]
        while c and c[-1] == 0:
            c.pop()
        self.c = c
        self.R = R

    def list(self): return [self.R.F(x) for x in self.c]
    def coefficients_int(self): return self.c
    def degree(self): return len(self.c) - 1
    def is_zero(self): return not self.c
    def constant_coefficient(self): return self.R.F(self.c[0] if self.c else 0)
    def leading_coefficient(self): return self.R.F(self.c[-1] if self.c else 0)

    def __call__(self, x):
        p = self.R.F.p
        x = int(x) % p
        acc = 0
        for a in reversed(self.c):
            acc = (acc * x + a) % p
        return self.R.F(acc)

    def _coerce(self, o):
        if isinstance(o, Polynomial):
            return o
        return Polynomial([int(o)], self.R)

    def __add__(self, o):
        o = self._coerce(o)
        a, b = (self.c, o.c) if len(self.c) >= len(o.c) else (o.c, self.c)
        r = list(a)
        for i, x in enumerate(b):
            r[i] += x
        return Polynomial(r, self.R)
    __radd__ = __add__

    def __neg__(self): return Polynomial([-x for x in self.c], self.R)
    def __sub__(self, o): return self + (-self._coerce(o))
    def __rsub__(self, o): return self._coerce(o) - self

    def __mul__(self, o):
        if not isinstance(o, Polynomial):
            s = int(o) % self.R.F.p
            return Polynomial([x * s for x in self.c], self.R)
        if not self.c or not o.c:
            return Polynomial([], self.R)
        r = [0] * (len(self.c) + len(o.c) - 1)
        for i, x in enumerate(self.c):
            if x:
                for j, y in enumerate(o.c):
                    r[i + j] += x * y
        return Polynomial(r, self.R)
    __rmul__ = __mul__

    def __pow__(self, e):
        r = Polynomial([1], self.R)
        b = self
        e = int(e)
        while e:
            if e & 1:
                r = r * b
            b = b * b
            e >>= 1
        return r

    def __divmod__(self, o):
        o = self._coerce(o)
        if not o.c:
            raise ZeroDivisionError("polynomial division by zero")
        p = self.R.F.p
        rem = list(self.c)
        dq = len(rem) - len(o.c)
        if dq < 0:
            return Polynomial([], self.R), self
        q = [0] * (dq + 1)
        inv = pow(o.c[-1], -1, p)
        for i in range(dq, -1, -1):
            f = rem[i + len(o.c) - 1] * inv % p
            q[i] = f
            if f:
                for j, y in enumerate(o.c):
                    rem[i + j] = (rem[i + j] - f * y) % p
        return Polynomial(q, self.R), Polynomial(rem[:len(o.c) - 1], self.R)

    def __floordiv__(self, o): return divmod(self, o)[0]
    def __mod__(self, o): return divmod(self, o)[1]
    def __truediv__(self, o):
        q, r = divmod(self, o)
        if r.c:
            raise ArithmeticError("inexact polynomial division")
        return q

    def __eq__(self, o):
        if isinstance(o, Polynomial):
            return self.c == o.c and self.R.F.p == o.R.F.p
        try:
            return self.c == Polynomial([int(o)], self.R).c
        except (TypeError, ValueError):
            return NotImplemented
    def __hash__(self): return hash((tuple(self.c), self.R.F.p))
    def __repr__(self):
        return "Polynomial(" + repr(self.c) + ")"


class PolynomialRing:
    """PolynomialRing(F, 'X') (kzg.py:53): R(list) / R(scalar) construct, R.gen() is X."""

    def __init__(self, F, name="X"):
        self.F = F
        self.name = name

    def __call__(self, x=0):
        if isinstance(x, Polynomial):
            return x
        if isinstance(x, (list, tuple)):
            return Polynomial(x, self)
        return Polynomial([int(x)], self)

    def gen(self): return Polynomial([0, 1], self)
    def base_ring(self): return self.F
    def __eq__(self, o): return isinstance(o, PolynomialRing) and o.F == self.F
    def __hash__(self): return hash(("R", self.F.p))


def field_modulus(F):
    """Modulus of a field object: ours or a Sage GF (both answer order())."""
    if hasattr(F, "order"):
        return int(F.order())
    if hasattr(F, "characteristic"):
        return int(F.characteristic())
    raise TypeError("cannot determine the modulus of field object %r" % (F,))
