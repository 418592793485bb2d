"""PLONK (GWC19, the version whose linearisation polynomial vanishes at zeta) over the engine's
KZG facade: the harness of BASELINE config 5 and SURVEY.md section 8f N2, at fixture scale.

The reference's plonk/{encoder,indexer,prover,verifier}.py are CALLERS of the hot path; this
module plays their role so that a full index -> prove -> verify round drives fft_ff_interpolation,
KZG.commit, KZG.open and KZG.batch_check exactly as the reference's round does: the same
transcript labels and order (plonk/prover.py:54-185, plonk/verifier.py:89-121), the same proof
and key dictionary shapes (plonk/prover.py:188-210, plonk/indexer.py:92-118), 4 INTTs and 9 MSMs
of size n+2..n+6 per proof plus two openings.

Polynomial algebra here is the host shim (schoolbook products): fine for the reference's 16-gate
fixture and the small synthetic circuits of the tests.  The n = 2^20 round runs in
plonk_device.py, where the quotient is computed on the device (coset NTTs of size 4n instead of
dense products); both provers share this module's Verifier."""
from .fft_ff import fft_ff_interpolation
from .kzg import KZG
from .transcript import Transcript


class Domain:
    """Multiplicative subgroup H of order n (a power of two) and the two coset shifts k1, k2
    (plonk/encoder.py:37-97).  The reference samples k1, k2 at random; any values with
    k1^n != 1, k2^n != 1, (k1/k2)^n != 1 serve, so the smallest suitable integers are used."""

    def __init__(self, Fq, n_gates):
        self.Fq = Fq
        self.n = 1 << max(0, (n_gates - 1).bit_length())
        self.g = Fq.root_of_unity(self.n)
        self._H = None                           # the n points themselves: built when first asked for (the verifier never does)
        n, ks = self.n, []
        cand = 2
        while len(ks) < 2:
            k = Fq(cand)
            if k ** n != 1 and all((k / o) ** n != 1 for o in ks):
                ks.append(k)
            cand += 1
        self.k1, self.k2 = ks

    @property
    def H(self):
        if self._H is None:
            H = [self.Fq(1)]
            for _ in range(self.n - 1):
                H.append(H[-1] * self.g)
            self._H = H
        return self._H

    def sigma_star(self, perm):
        """Position j in [0, 3n) -> its label in H, k1*H or k2*H; composed with the permutation."""
        n = self.n
        label = self.H + [self.k1 * h for h in self.H] + [self.k2 * h for h in self.H]
        return [label[perm[i]] for i in range(3 * n)]

    def interpolate(self, values):
        return fft_ff_interpolation(list(values), self.g, self.Fq)

    def lagrange_1_at(self, x):
        n = self.n
        return (x ** n - 1) / (self.Fq(n) * (x - 1))

    def public_input_at(self, x, zeta):
        """PI(zeta) = -sum_i x_i L_i(zeta) with L_i(X) = g^i (X^n - 1) / (n (X - g^i)): what the verifier needs
        (plonk/verifier.py evaluates the interpolated PI polynomial; same value), in O(len(x)) field operations
        instead of an n-point interpolation."""
        Fq, n = self.Fq, self.n
        zeta = Fq(zeta)
        zh = zeta ** n - 1
        acc, gi = Fq(0), Fq(1)
        for v in x:
            if zeta == gi:                       # zeta on the domain: L_i(zeta) = 1, every other basis polynomial vanishes
                return -Fq(v)
            acc += Fq(v) * gi * zh / (Fq(n) * (zeta - gi))
            gi *= self.g
        return -acc

    def public_input_poly(self, R, x):
        """PI(X) = -sum_i x_i L_i(X) over the first len(x) rows."""
        vals = [-self.Fq(v) for v in x] + [self.Fq(0)] * (self.n - len(x))
        return self.interpolate(vals)


def pad(values, n, Fq):
    return [Fq(v) for v in values] + [Fq(0)] * (n - len(values))


class Indexer:
    def __init__(self, curve_type="bn254"):
        self.kzg = KZG(curve_type)

    def preprocess(self, qM, qL, qR, qO, qC, perm, max_degree=None, tau=None):
        kzg, Fq = self.kzg, self.kzg.Fq
        dom = Domain(Fq, len(qM))
        n = dom.n
        if len(perm) != 3 * n:
            # wires of padding rows map to themselves
            m = len(qM)
            full = list(range(3 * n))
            for blk in range(3):
                for i in range(m):
                    j = perm[blk * m + i]
                    full[blk * n + i] = (j // m) * n + (j % m)
            perm = full
        ck, rk = kzg.setup(max_degree if max_degree is not None else n + 5, tau=tau)     # main.py:85
        sel = {name: dom.interpolate(pad(v, n, Fq)) for name, v in
               (("qM", qM), ("qL", qL), ("qR", qR), ("qO", qO), ("qC", qC))}
        sstar = dom.sigma_star(perm)
        sig = {"S_sigma1": dom.interpolate(sstar[:n]), "S_sigma2": dom.interpolate(sstar[n:2 * n]),
               "S_sigma3": dom.interpolate(sstar[2 * n:])}
        polys = {**sel, **sig}
        names = list(polys)
        comms = dict(zip(names, kzg.commit(ck, [polys[k] for k in names])))        # 8 MSMs (plonk/indexer.py:77)
        sub = {"n": n, "g": dom.g, "k1": dom.k1, "k2": dom.k2, "H": dom.H}
        ipk = {"ck": ck, "polynomials": polys, "subgroups": sub, "sigma_star": sstar,
               "vanishing_poly": kzg.X ** n - 1, "commitments": comms}
        ivk = {"rk": rk, "commitments": comms, "subgroups": sub}
        return ipk, ivk


class Prover:
    def __init__(self, curve_type="bn254", sharding=None):
        """`sharding`: a sharding.ProofSharding when one proof is produced by several ranks (BASELINE config 5 on
        more than one GPU): the commitments of a round and the two openings are dealt over the ranks against a
        replicated key, and every rank returns the same proof."""
        self.kzg = KZG(curve_type)
        self.sharding = sharding

    def prove(self, ipk, x, w, blinders=None, trace=None):
        """`blinders` (tests only) fixes b1..b11 of plonk/prover.py:72-75 and :346; `trace`, when a
        dict, receives the challenges and every intermediate polynomial."""
        kzg, Fq, R, X = self.kzg, self.kzg.Fq, self.kzg.R, self.kzg.X
        ck, P = ipk["ck"], ipk["polynomials"]
        sub = ipk["subgroups"]
        n, g, k1, k2, H = sub["n"], sub["g"], sub["k1"], sub["k2"], sub["H"]
        ZH = ipk["vanishing_poly"]
        sstar = ipk["sigma_star"]
        dom = Domain(Fq, n)
        tr = Transcript("plonk-proof", Fq)
        tr.append_message("public-inputs", x)
        full = [Fq(v) for v in list(x) + list(w)]
        m = len(full) // 3
        cols = [pad(full[i * m:(i + 1) * m], n, Fq) for i in range(3)]
        PI = dom.public_input_poly(R, x)
        b = [Fq.random_element() for _ in range(11)] if blinders is None else [Fq(v) for v in blinders]
        assert len(b) == 11
        sh = self.sharding
        if sh is not None and sh.active:
            b = [Fq(v) for v in sh.shared_scalars([int(v) for v in b])]              # drawn once, by rank 0

        def commit(polys):                                                          # plonk/prover.py:89,113,136
            return kzg.commit(ck, polys) if sh is None else sh.commit_batch(lambda ps: kzg.commit(ck, ps), polys)

        # round 1: wire polynomials (3 INTTs, 3 MSMs of degree n+1)
        wires = [(b[2 * i] * X + b[2 * i + 1]) * ZH + dom.interpolate(cols[i]) for i in range(3)]
        a_p, b_p, c_p = wires
        wire_comms = commit(wires)
        tr.append_message("round1-commitments", wire_comms)
        beta, gamma = tr.get_challenge("beta"), tr.get_challenge("gamma")

        # round 2: permutation accumulator (1 INTT, 1 MSM of degree n+2)
        acc = [Fq(1)]
        for i in range(n - 1):
            num = den = Fq(1)
            for col, shift, blk in ((cols[0], Fq(1), 0), (cols[1], k1, 1), (cols[2], k2, 2)):
                num *= col[i] + beta * shift * H[i] + gamma
                den *= col[i] + beta * sstar[blk * n + i] + gamma
            acc.append(acc[-1] * num / den)
        z_p = (b[6] * X * X + b[7] * X + b[8]) * ZH + dom.interpolate(acc)
        z_comm = commit([z_p])[0]
        tr.append_message("round2-commitment", z_comm)
        alpha = tr.get_challenge("alpha")

        # round 3: quotient (dense products on the host at this scale), 3 MSMs of degree <= n+5
        zw_p = R([c * g ** i for i, c in enumerate(z_p.list())])                    # z(X*g)
        L1 = (X ** n - 1) // ((X - 1) * Fq(n))
        gate = a_p * b_p * P["qM"] + a_p * P["qL"] + b_p * P["qR"] + c_p * P["qO"] + PI + P["qC"]
        perm1 = (a_p + beta * X + gamma) * (b_p + beta * k1 * X + gamma) * (c_p + beta * k2 * X + gamma) * z_p
        perm2 = ((a_p + beta * P["S_sigma1"] + gamma) * (b_p + beta * P["S_sigma2"] + gamma)
                 * (c_p + beta * P["S_sigma3"] + gamma) * zw_p)
        numer = gate + alpha * (perm1 - perm2) + alpha * alpha * ((z_p - 1) * L1)
        t_p, rem = divmod(numer, ZH)
        assert rem.is_zero(), "constraint system is not satisfied"
        tc = t_p.list() + [Fq(0)] * (3 * n + 6)
        t_lo = R(tc[:n]) + b[9] * X ** n
        t_mid = R(tc[n:2 * n]) - b[9] + b[10] * X ** n
        t_hi = R(tc[2 * n:3 * n + 6]) - b[10]
        t_comms = commit([t_lo, t_mid, t_hi])
        tr.append_message("round3-commitments", t_comms)
        zeta = tr.get_challenge("zeta")

        # round 4: evaluations
        ev = {"a": a_p(zeta), "b": b_p(zeta), "c": c_p(zeta), "s_sigma1": P["S_sigma1"](zeta),
              "s_sigma2": P["S_sigma2"](zeta), "z_omega": z_p(zeta * g)}
        tr.append_message("round4-evaluations", [ev[k] for k in ("a", "b", "c", "s_sigma1", "s_sigma2", "z_omega")])
        v = tr.get_challenge("v")

        # round 5: linearisation polynomial r (r(zeta) = 0) and the two openings
        za, zb, zc, s1, s2, zo = ev["a"], ev["b"], ev["c"], ev["s_sigma1"], ev["s_sigma2"], ev["z_omega"]
        zn = zeta ** n
        r_p = (za * zb * P["qM"] + za * P["qL"] + zb * P["qR"] + zc * P["qO"] + PI(zeta) + P["qC"]
               + alpha * ((za + beta * zeta + gamma) * (zb + beta * k1 * zeta + gamma) * (zc + beta * k2 * zeta + gamma) * z_p
                          - (za + beta * s1 + gamma) * (zb + beta * s2 + gamma) * zo * (zc + beta * P["S_sigma3"] + gamma))
               + alpha * alpha * dom.lagrange_1_at(zeta) * (z_p - 1)
               - (zn - 1) * (t_lo + zn * t_mid + zn * zn * t_hi))
        assert r_p(zeta) == 0, "r(zeta) should be zero"                             # plonk/prover.py:171
        if trace is not None:
            trace.update(beta=beta, gamma=gamma, alpha=alpha, zeta=zeta, v=v, evaluations=dict(ev), a=a_p, b=b_p,
                         c=c_p, z=z_p, PI=PI, t=t_p, t_lo=t_lo, t_mid=t_mid, t_hi=t_hi, r=r_p)
        opens = [lambda: kzg.open(ck, [r_p, a_p, b_p, c_p, P["S_sigma1"], P["S_sigma2"]], zeta, v),   # plonk/prover.py:184
                 lambda: kzg.open(ck, [z_p], zeta * g, v)]                                               # :185
        W_z, W_zw = [f() for f in opens] if sh is None else sh.run_dealt(opens)
        return {"commitments": dict(zip(("a", "b", "c"), wire_comms), z=z_comm,
                                    t_lo=t_comms[0], t_mid=t_comms[1], t_hi=t_comms[2]),
                "evaluations": ev,
                "kzg_proofs": {"W_z": W_z, "W_zw": W_zw}}


class Verifier:
    def __init__(self, curve_type="bn254"):
        self.kzg = KZG(curve_type)

    def verify(self, ivk, x, proof):
        kzg, Fq, R = self.kzg, self.kzg.Fq, self.kzg.R
        sub, VC = ivk["subgroups"], ivk["commitments"]
        n, g, k1, k2 = sub["n"], sub["g"], sub["k1"], sub["k2"]
        C, ev, W = proof["commitments"], proof["evaluations"], proof["kzg_proofs"]
        dom = Domain(Fq, n)
        tr = Transcript("plonk-proof", Fq)
        tr.append_message("public-inputs", x)
        tr.append_message("round1-commitments", [C["a"], C["b"], C["c"]])
        beta, gamma = tr.get_challenge("beta"), tr.get_challenge("gamma")
        tr.append_message("round2-commitment", C["z"])
        alpha = tr.get_challenge("alpha")
        tr.append_message("round3-commitments", [C["t_lo"], C["t_mid"], C["t_hi"]])
        zeta = tr.get_challenge("zeta")
        za, zb, zc = Fq(ev["a"]), Fq(ev["b"]), Fq(ev["c"])
        s1, s2, zo = Fq(ev["s_sigma1"]), Fq(ev["s_sigma2"]), Fq(ev["z_omega"])
        tr.append_message("round4-evaluations", [za, zb, zc, s1, s2, zo])
        v = tr.get_challenge("v")
        u = tr.get_challenge("u")
        zn = zeta ** n
        L1z = dom.lagrange_1_at(zeta)
        PIz = dom.public_input_at(x, zeta)
        mul, add, neg, G1 = kzg.multiply, kzg.add, kzg.neg, kzg.G1

        def lin(terms):
            acc = kzg.Z1
            for pt, s in terms:
                acc = add(acc, mul(pt, int(Fq(s))))
            return acc

        f1 = (za + beta * zeta + gamma) * (zb + beta * k1 * zeta + gamma) * (zc + beta * k2 * zeta + gamma)
        f2 = (za + beta * s1 + gamma) * (zb + beta * s2 + gamma) * zo
        r_comm = lin([
            (VC["qM"], za * zb), (VC["qL"], za), (VC["qR"], zb), (VC["qO"], zc), (VC["qC"], 1), (G1, PIz),
            (C["z"], alpha * f1 + alpha * alpha * L1z),
            (VC["S_sigma3"], -alpha * f2 * beta),
            (G1, -alpha * f2 * (zc + gamma) - alpha * alpha * L1z),
            (C["t_lo"], -(zn - 1)), (C["t_mid"], -(zn - 1) * zn), (C["t_hi"], -(zn - 1) * zn * zn),
        ])
        return kzg.batch_check(
            ivk["rk"],
            [[r_comm, C["a"], C["b"], C["c"], VC["S_sigma1"], VC["S_sigma2"]], [C["z"]]],
            [zeta, zeta * g],
            [[Fq(0), za, zb, zc, s1, s2], [zo]],
            [W["W_z"], W["W_zw"]],
            [v, v], u)


def synthetic_circuit(n_gates, Fq, seed=1):
    """A mul/add chain in the shape of the reference's fixture: rows 0..k-1
    expose public inputs, then alternating multiplication and addition gates chained by copy
    constraints, padded with empty rows.  Returns (qM, qL, qR, qO, qC, perm, x, w)."""
    import random
    rng = random.Random(seed)
    n = n_gates
    k = min(4, max(1, n // 4))
    pub = [rng.randrange(2, 50) for _ in range(k)]
    qM, qL, qR, qO, qC = ([0] * n for _ in range(5))
    a, b, c = [0] * n, [0] * n, [0] * n
    uses = {}                                   # value id -> list of wire positions holding it

    def put(col, row, val, vid):
        (a, b, c)[col][row] = val
        uses.setdefault(vid, []).append(col * n + row)

    vals = {}
    for i in range(k):                          # public-input rows: qL = 1, a = c = x_i, PI forces the value
        qL[i] = 1
        vals[i] = pub[i]
        put(0, i, pub[i], i)
        put(2, i, pub[i], ("pubc", i))
    last, nid = 0, k
    for row in range(k, n - 1):
        other = rng.randrange(0, nid)
        x1, x2 = vals[last], vals[other]
        if (row - k) % 2 == 0:
            qM[row], qO[row], out = 1, -1, x1 * x2
        else:
            qL[row], qR[row], qO[row], out = 1, 1, -1, x1 + x2
        out %= Fq.p
        put(0, row, x1, last)
        put(1, row, x2, other)
        vals[nid] = out
        put(2, row, out, nid)
        last, nid = nid, nid + 1
    perm = list(range(3 * n))
    for pos in uses.values():                   # one cycle per value
        for i, p in enumerate(pos):
            perm[p] = pos[(i + 1) % len(pos)]
    w_full = a + b + c
    return qM, qL, qR, qO, qC, perm, w_full[:k], w_full[k:]
