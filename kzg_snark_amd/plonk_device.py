"""PLONK prover with every polynomial resident in HBM -- BASELINE config 5 ("full plonk/prover.py
round at n = 2^20 gates, all commits/NTTs on the GPU"), SURVEY.md section 8f N2.

Same protocol, transcript and proof dictionary as kzg_snark_amd/plonk.py (and therefore the same
Verifier); what changes is where the algebra runs.  The reference does the prover's polynomial
algebra with Sage's dense polynomials on the host (plonk/prover.py:243-316: n-1 sequential field
divisions for the accumulator, degree-3n products and a division by X^n - 1 for the quotient).
Here:
  * wire / accumulator polynomials: device INTTs (kzg_ntt_device);
  * accumulator: ratios by batch inversion + exclusive prefix product (kzg_fr_vec_inverse,
    kzg_fr_vec_prefix_product) instead of a sequential loop;
  * quotient: every polynomial is evaluated on the coset K*H' of the size-4n subgroup (coefficient
    shift + 4n-point NTT), the constraint is combined pointwise, divided by Z_H (which takes four
    values on the coset) and interpolated back -- no dense products, no polynomial division;
  * evaluations at zeta: kzg_fr_poly_eval; linearisation: kzg_fr_vec_lincomb;
  * commitments / openings: kzg_commit_device / kzg_open_device.
Vectors are int64[n, 4] torch tensors (canonical 32-byte Fr elements)."""
import numpy as np
import torch

from . import _native
from .kzg import KZG, CommitmentKey
from .plonk import Domain
from .transcript import Transcript


class DeviceAlgebra:
    """Thin helper around the C ABI's kzg_fr_* and NTT entry points for torch tensors."""

    def __init__(self, curve_type, device=None):
        """`device`: "cuda:N" / N / None = the process's GPU (_native.default_device(): one process per GPU)."""
        if device is None:
            idx = _native.default_device()
        elif isinstance(device, int):
            idx = device
        else:
            idx = torch.device(device).index or 0
        self.ctx = _native.get_context(curve_type, idx)
        self.dev = torch.device("cuda", idx)
        self.stream = torch.cuda.Stream(device=self.dev)
        torch.cuda.set_stream(self.stream)        # every torch op of this thread shares the library's stream
        self.ctx.set_stream(self.stream.cuda_stream)
        self.r = KZG(curve_type).curve_order

    # ---- construction
    def upload(self, values):
        arr = _native.ints_to_limbs([int(v) % self.r for v in values])
        with torch.cuda.stream(self.stream):
            return torch.from_numpy(arr.view(np.int64)).to(self.dev, non_blocking=False)

    def upload_limbs(self, arr):
        """uint64[n, 4] canonical little-endian limbs (what a witness generator would emit) -> device."""
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        with torch.cuda.stream(self.stream):
            return torch.from_numpy(arr.view(np.int64)).to(self.dev, non_blocking=False)

    def upload_parts(self, n, parts):
        """Concatenation of uint64[k_j, 4] arrays (sum k_j = n) as one device vector."""
        out = torch.empty((n, 4), dtype=torch.int64, device=self.dev)
        at = 0
        with torch.cuda.stream(self.stream):
            for a in parts:
                a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
                out[at:at + a.shape[0]].copy_(torch.from_numpy(a.view(np.int64)))
                at += a.shape[0]
        assert at == n
        return out

    def download(self, t, count=None):
        self.ctx.synchronize()
        a = t[:count] if count is not None else t
        return _native.limbs_to_ints(a.cpu().numpy().view(np.uint64))

    def zeros(self, n):
        with torch.cuda.stream(self.stream):
            return torch.zeros((n, 4), dtype=torch.int64, device=self.dev)

    def const(self, n, value):
        with torch.cuda.stream(self.stream):
            row = torch.from_numpy(_native.int_to_words(int(value) % self.r).view(np.int64)).to(self.dev)
            return row.repeat(n, 1).contiguous()

    def padded(self, t, n):
        with torch.cuda.stream(self.stream):
            out = torch.zeros((n, 4), dtype=torch.int64, device=self.dev)
            out[:t.shape[0]] = t
            return out

    # ---- element-wise
    def op(self, kind, a, b):
        out = torch.empty_like(a)
        self.ctx.vec_op(kind, a.shape[0], a.data_ptr(), b.data_ptr(), out.data_ptr())
        return out

    def mul(self, a, b): return self.op("mul", a, b)
    def add(self, a, b): return self.op("add", a, b)
    def sub(self, a, b): return self.op("sub", a, b)

    def lincomb(self, n, terms):
        """sum_j s_j * v_j over the first n entries; terms = [(scalar, tensor), ...]."""
        out = torch.empty((n, 4), dtype=torch.int64, device=self.dev)
        self.ctx.vec_lincomb(n, [t.data_ptr() for _, t in terms], [min(t.shape[0], n) for _, t in terms],
                             [int(s) % self.r for s, _ in terms], out.data_ptr())
        return out

    def mul_powers(self, a, s, c0=1):
        out = torch.empty_like(a)
        self.ctx.vec_mul_powers(a.shape[0], a.data_ptr(), int(s) % self.r, int(c0) % self.r, out.data_ptr())
        return out

    def inverse(self, a):
        out = torch.empty_like(a)
        self.ctx.vec_inverse(a.shape[0], a.data_ptr(), out.data_ptr())
        return out

    def prefix_product(self, a):
        out = torch.empty_like(a)
        self.ctx.vec_prefix_product(a.shape[0], a.data_ptr(), out.data_ptr())
        return out

    def eval(self, coeffs, z):
        return self.ctx.poly_eval(coeffs.shape[0], coeffs.data_ptr(), int(z) % self.r)

    def ntt(self, t, w, inverse):
        n = t.shape[0]
        self.ctx.ntt_device(t.data_ptr(), n.bit_length() - 1, _native.int_to_words(int(w) % self.r), inverse, 1)
        return t

    def set_entries(self, t, updates):
        """t[i] = (t[i] + delta) mod r for a handful of (i, delta) pairs (blinding terms), on the device: the
        deltas travel in one pinned, stream-ordered copy and each entry is a one-element field addition, so the
        host never waits for the work queued before it."""
        if not updates:
            return
        rows = np.stack([_native.int_to_words(int(d) % self.r) for _, d in updates]).view(np.int64)
        host = torch.from_numpy(rows).pin_memory()
        with torch.cuda.stream(self.stream):
            dl = host.to(self.dev, non_blocking=True)          # the pinned block is recycled only after the copy ran
        for j, (i, _) in enumerate(updates):
            assert 0 <= i < t.shape[0]
            p = t.data_ptr() + 32 * i
            self.ctx.vec_op("add", 1, p, dl.data_ptr() + 32 * j, p)


class DeviceIndexer:
    def __init__(self, curve_type="bls12_381", device=None):
        self.kzg = KZG(curve_type)
        self.alg = DeviceAlgebra(curve_type, device)

    def preprocess(self, qM, qL, qR, qO, qC, perm, tau=None):
        kzg, Fq, alg = self.kzg, self.kzg.Fq, self.alg
        dom = Domain(Fq, len(qM))
        n = dom.n
        assert len(qM) == n and len(perm) == 3 * n, "pad the circuit to a power of two first"
        ck, rk = kzg.setup(n + 5, tau=tau)                                           # main.py:85
        # sigma*: position j of [0, 3n) carries the label of position perm[j], the labels being H, k1 H, k2 H
        # (plonk/encoder.py:160-194) -- built and permuted on the device (3n field elements never exist as Python ints)
        ones = alg.const(n, 1)
        labels = torch.cat([alg.mul_powers(ones, dom.g, 1), alg.mul_powers(ones, dom.g, int(dom.k1)),
                            alg.mul_powers(ones, dom.g, int(dom.k2))])
        with torch.cuda.stream(alg.stream):
            perm_t = torch.as_tensor(np.asarray(perm, dtype=np.int64)).to(alg.dev)
            assert perm_t.shape[0] == 3 * n
            sstar = labels.index_select(0, perm_t)
        sigma_values = {f"S_sigma{b + 1}": sstar[b * n:(b + 1) * n].contiguous() for b in range(3)}
        cols = {"qM": qM, "qL": qL, "qR": qR, "qO": qO, "qC": qC}
        on_dev = {k: alg.upload(v) for k, v in cols.items()}
        on_dev.update({k: v.clone() for k, v in sigma_values.items()})
        coeffs = {k: alg.ntt(v, dom.g, True) for k, v in on_dev.items()}               # 8 INTTs
        pack = torch.stack([coeffs[k] for k in coeffs]).contiguous()
        xy, inf = alg.ctx.commit_device(ck.srs, pack.data_ptr(), [n] * len(coeffs), n)  # 8 MSMs
        comms = dict(zip(coeffs, kzg._points(xy, inf)))
        sub = {"n": n, "g": dom.g, "k1": dom.k1, "k2": dom.k2}
        ipk = {"ck": ck, "coeffs": coeffs, "sigma_values": sigma_values,
               "subgroups": sub, "commitments": comms}
        ivk = {"rk": rk, "commitments": comms, "subgroups": sub}
        return ipk, ivk


class DeviceProver:
    def __init__(self, curve_type="bls12_381", alg=None, sharding=None):
        """`sharding`: a sharding.ProofSharding when one proof is produced by several ranks, one GPU each
        (BASELINE config 5 on more than one GPU): every rank runs the vector work of the rounds on its own copy
        of the polynomials, the MSMs of a round -- 3 + 1 + 3 commitments (plonk/prover.py:89,113,136) and the two
        openings (:184-185) -- are dealt round-robin against the replicated key, and one all-gather of 97-byte
        records per round gives every rank the same points, hence the same challenges and the same proof."""
        self.kzg = KZG(curve_type)
        self.alg = alg or DeviceAlgebra(curve_type)
        self.sharding = sharding
        self._dom_cache = {}                     # (n, g) -> device vectors of that evaluation domain (one entry)

    def _domain_constants(self, n, g):
        """Vectors that depend only on the evaluation domain (not on the circuit or the witness), built once per
        domain size and kept on the device: 1, g^i on H; on the coset K*H' of the size-4n subgroup the points x,
        1/Z_H(x) and L1(x) = Z_H(x) / (n (x - 1))."""
        key = (n, int(g))
        cache = self._dom_cache
        if key not in cache:
            alg, Fq = self.alg, self.kzg.Fq
            r = self.kzg.curve_order
            N4 = 4 * n
            w4, K = int(Fq.root_of_unity(N4)), int(Fq.multiplicative_generator())
            ones = alg.const(n, 1)
            ones4 = alg.const(N4, 1)
            xs = alg.mul_powers(ones4, w4, K)                                     # the coset points
            xn = alg.mul_powers(ones4, pow(w4, n, r), pow(K, n, r))               # x^n: four distinct values
            zh = alg.sub(xn, ones4)
            cache.clear()                                                          # one domain at a time (N4-sized vectors)
            cache[key] = {"ones": ones, "idH": alg.mul_powers(ones, g), "ones4": ones4, "xs": xs,
                          "zh_inv": alg.inverse(zh), "w4": w4, "K": K, "e0": alg.const(1, 1),
                          "l1": alg.mul(zh, alg.inverse(alg.lincomb(N4, [(n, xs), (-n, ones4)])))}
        return cache[key]

    def _circuit_cosets(self, ipk, on_coset):
        """Coset evaluations of the eight preprocessed polynomials: fixed per circuit, so they live with the
        proving key (4n elements each) instead of being re-transformed for every proof."""
        cache = ipk.get("_coset_evals")
        if cache is None:
            cache = ipk["_coset_evals"] = {k: on_coset(ipk["coeffs"][k]) for k in
                                           ("qM", "qL", "qR", "qO", "qC", "S_sigma1", "S_sigma2", "S_sigma3")}
        return cache

    def _commit(self, ck, tensors):
        return self._commit_end(self._commit_begin(ck, tensors)), None

    def _commit_begin(self, ck, tensors):
        """Queue the commitments of a round on the library's pipeline and return at once: whatever the prover
        enqueues next that does not need the round's challenge runs beside the MSMs (they are instruction-issue
        bound, the vector work is HBM-bound).  With several ranks only this rank's share of the round is queued."""
        alg, sh = self.alg, self.sharding
        k = len(tensors)
        mine = list(range(k)) if sh is None else sh.mine(k)
        if not mine:
            return None, None, None, mine, k
        own = [tensors[i] for i in mine]
        stride = max(t.shape[0] for t in own)
        pack = torch.stack([alg.padded(t, stride) for t in own]).contiguous()
        xy = np.zeros((len(own), 2 * alg.ctx.fp_limbs), dtype=np.uint64)
        inf = np.zeros(len(own), dtype=np.uint8)
        alg.ctx.commit_device_async(ck.srs, pack.data_ptr(), [t.shape[0] for t in own], stride, xy, inf)
        return pack, xy, inf, mine, k

    def _commit_end(self, handle):
        _, xy, inf, mine, k = handle
        self.alg.ctx.commit_flush()
        local = self.kzg._points(xy, inf) if mine else []
        if self.sharding is None:
            return local
        return self.sharding.gather_points(dict(zip(mine, local)), k)      # the round's one exchange

    def _blind(self, coeffs, n, blinders):
        """coeffs (length n) + (b_k X^k + ... + b_0) * (X^n - 1): length n + len(blinders)."""
        alg = self.alg
        out = alg.padded(coeffs, n + len(blinders))
        ups = []
        for k, bk in enumerate(blinders):            # blinders[k] multiplies X^k
            ups.append((k, -int(bk)))
            ups.append((n + k, int(bk)))
        alg.set_entries(out, ups)
        return out

    def prove(self, ipk, x, w, blinders=None, trace=None):
        """plonk/prover.py:24-212.  `blinders` (tests only) fixes b1..b11 of plonk/prover.py:72-75 and
        :346 in the reference's order; `trace`, when a dict, receives the challenges and the device
        tensors of every intermediate polynomial so a test can compare them with the oracle."""
        try:
            return self._prove(ipk, x, w, blinders, trace)
        except BaseException:
            # a proof that fails half way (an assert of the protocol, an allocation) must not leave commitments of
            # its rounds queued in the library's pipeline: drain it before the error travels on
            try:
                self.alg.ctx.commit_flush()
            except Exception:
                pass
            raise

    def _prove(self, ipk, x, w, blinders, trace):
        kzg, Fq, alg = self.kzg, self.kzg.Fq, self.alg
        sh = self.sharding
        r = kzg.curve_order
        ck, C = ipk["ck"], ipk["coeffs"]
        sub = ipk["subgroups"]
        n, g, k1, k2 = sub["n"], sub["g"], sub["k1"], sub["k2"]
        assert isinstance(ck, CommitmentKey)
        dom = Domain.__new__(Domain)                 # reuse lagrange_1_at without rebuilding H
        dom.Fq, dom.n, dom.g = Fq, n, g
        tr = Transcript("plonk-proof", Fq)
        tr.append_message("public-inputs", x)
        x_limbs = _native.ints_to_limbs([int(v) % r for v in x]).reshape(-1, 4)
        if isinstance(w, np.ndarray):                # witness already in limb form: no per-element Python work
            w_limbs = np.ascontiguousarray(w, dtype=np.uint64).reshape(-1, 4)
        else:
            w_limbs = _native.ints_to_limbs([int(v) % r for v in w]).reshape(-1, 4)
        nx = x_limbs.shape[0]
        assert nx + w_limbs.shape[0] == 3 * n

        def column(i):
            """rows [i n, (i+1) n) of x ++ w, uploaded piecewise (no host-side copy of the 96 n bytes)."""
            lo, hi = i * n, (i + 1) * n
            parts = []
            if lo < nx:
                parts.append(x_limbs[lo:min(hi, nx)])
            if hi > nx:
                parts.append(w_limbs[max(lo, nx) - nx:hi - nx])
            return alg.upload_parts(n, parts)
        b = [int(Fq.random_element()) for _ in range(11)] if blinders is None else [int(v) % r for v in blinders]
        assert len(b) == 11
        if sh is not None and sh.active:
            b = sh.shared_scalars(b)                                          # drawn once, by rank 0
        dealt = sh is not None and sh.active and sh.deal_transforms
        D = self._domain_constants(n, g)
        ones, idH = D["ones"], D["idH"]                                       # 1 and g^i on H

        # round 1
        vals = [column(i) for i in range(3)]
        pi = alg.zeros(n)                                                    # PI values: -x_i on the first rows
        if len(x):
            pi[:len(x)] = alg.upload([(-int(v)) % r for v in x])
        if dealt:   # the four independent INTTs of the round, each on its owner, results broadcast
            srcs = vals + [pi]
            co = sh.dealt_tensors([(n, 4)] * 4, lambda i: alg.ntt(srcs[i].clone(), g, True), vals[0])
        else:
            co = [alg.ntt(v.clone(), g, True) for v in vals] + [None]
        wires = [self._blind(co[i], n, [b[2 * i + 1], b[2 * i]]) for i in range(3)]
        a_c, b_c, c_c = wires
        h1 = self._commit_begin(ck, wires)
        # queued behind the round-1 MSMs, needing no challenge: the wires and PI on the coset K * H', |H'| = 4n,
        # and the gate constraint
        N4 = 4 * n
        w4, K = D["w4"], D["K"]

        def on_coset(coeffs):
            return alg.ntt(alg.mul_powers(alg.padded(coeffs, N4), K), w4, False)

        PI_c = co[3] if dealt else alg.ntt(pi, g, True)
        r1 = (("a", a_c), ("b", b_c), ("c", c_c), ("PI", PI_c))
        if dealt:   # and their four coset NTTs
            ev4 = sh.dealt_tensors([(N4, 4)] * 4, lambda i: on_coset(r1[i][1]), vals[0])
            E = {k: t for (k, _), t in zip(r1, ev4)}
        else:
            E = {k: on_coset(v) for k, v in r1}
        E.update(self._circuit_cosets(ipk, on_coset))
        gate = alg.add(alg.add(alg.mul(alg.mul(E["a"], E["b"]), E["qM"]), alg.mul(E["a"], E["qL"])),
                       alg.add(alg.mul(E["b"], E["qR"]), alg.mul(E["c"], E["qO"])))
        gate = alg.add(gate, alg.add(E["PI"], E["qC"]))
        wire_comms = self._commit_end(h1)
        tr.append_message("round1-commitments", wire_comms)
        beta, gamma = int(tr.get_challenge("beta")), int(tr.get_challenge("gamma"))

        # round 2: z_i = prod_{j<i} num_j / den_j
        S = ipk["sigma_values"]
        num = den = None
        for v, shift, sig in ((vals[0], 1, S["S_sigma1"]), (vals[1], int(k1), S["S_sigma2"]), (vals[2], int(k2), S["S_sigma3"])):
            fn = alg.lincomb(n, [(1, v), (beta * shift, idH), (gamma, ones)])
            fd = alg.lincomb(n, [(1, v), (beta, sig), (gamma, ones)])
            num = fn if num is None else alg.mul(num, fn)
            den = fd if den is None else alg.mul(den, fd)
        z_vals = alg.prefix_product(alg.mul(num, alg.inverse(den)))
        z_c = self._blind(alg.ntt(z_vals, g, True), n, [b[8], b[7], b[6]])
        h2 = self._commit_begin(ck, [z_c])
        # queued behind the round-2 MSM: everything of the quotient that needs beta and gamma but not alpha
        E["z"] = on_coset(z_c)
        ones4, xs = D["ones4"], D["xs"]                                       # 1 and the coset points
        zw = torch.roll(E["z"], shifts=-4, dims=0).contiguous()              # z(g * x): g = w4^4
        p1 = p2 = None
        for key, shift, sig in (("a", 1, "S_sigma1"), ("b", int(k1), "S_sigma2"), ("c", int(k2), "S_sigma3")):
            f1 = alg.lincomb(N4, [(1, E[key]), (beta * shift, xs), (gamma, ones4)])
            f2 = alg.lincomb(N4, [(1, E[key]), (beta, E[sig]), (gamma, ones4)])
            p1 = f1 if p1 is None else alg.mul(p1, f1)
            p2 = f2 if p2 is None else alg.mul(p2, f2)
        perm = alg.sub(alg.mul(p1, E["z"]), alg.mul(p2, zw))
        # Z_H(x) = x^n - 1 and L1(x) = Z_H(x) / (n (x - 1)) on the coset: domain constants
        l1t = alg.mul(alg.sub(E["z"], ones4), D["l1"])
        z_comm = self._commit_end(h2)[0]
        tr.append_message("round2-commitment", z_comm)
        alpha = int(tr.get_challenge("alpha"))

        # round 3: combine with alpha, divide by Z_H, back to coefficients
        numer = alg.lincomb(N4, [(1, gate), (alpha, perm), (alpha * alpha, l1t)])
        t_ev = alg.mul(numer, D["zh_inv"])
        t_c = alg.mul_powers(alg.ntt(t_ev, w4, True), pow(K, -1, r))          # back to coefficients
        # the quotient must have degree <= 3n + 5: its tail is fetched behind the work queued so far and
        # looked at once the round's commitments have come back (no extra wait)
        tail = t_c[3 * n + 6:3 * n + 6 + 64]
        tail_host = torch.empty(tuple(tail.shape), dtype=torch.int64).pin_memory()
        with torch.cuda.stream(alg.stream):
            tail_host.copy_(tail, non_blocking=True)
        t_lo = alg.padded(t_c[:n], n + 1)
        t_mid = alg.padded(t_c[n:2 * n], n + 1)
        t_hi = t_c[2 * n:3 * n + 6].clone()
        alg.set_entries(t_lo, [(n, b[9])])
        alg.set_entries(t_mid, [(0, -b[9]), (n, b[10])])
        alg.set_entries(t_hi, [(0, -b[10])])
        t_comms, _ = self._commit(ck, [t_lo, t_mid, t_hi])
        assert not bool(tail_host.any()), "constraint system is not satisfied (quotient has a remainder)"
        tr.append_message("round3-commitments", t_comms)
        zeta = int(tr.get_challenge("zeta"))

        # round 4
        ev = {"a": alg.eval(a_c, zeta), "b": alg.eval(b_c, zeta), "c": alg.eval(c_c, zeta),
              "s_sigma1": alg.eval(C["S_sigma1"], zeta), "s_sigma2": alg.eval(C["S_sigma2"], zeta),
              "z_omega": alg.eval(z_c, zeta * int(g) % r)}
        evF = {k: Fq(v) for k, v in ev.items()}
        tr.append_message("round4-evaluations", [evF[k] for k in ("a", "b", "c", "s_sigma1", "s_sigma2", "z_omega")])
        v = int(tr.get_challenge("v"))

        # round 5: r(X) as a scalar combination of coefficient vectors
        za, zb, zc, s1, s2, zo = (ev[k] for k in ("a", "b", "c", "s_sigma1", "s_sigma2", "z_omega"))
        zn = pow(zeta, n, r)
        L1z = int(dom.lagrange_1_at(Fq(zeta)))
        PIz = alg.eval(PI_c, zeta)
        f1 = (za + beta * zeta + gamma) * (zb + beta * int(k1) * zeta + gamma) * (zc + beta * int(k2) * zeta + gamma) % r
        f2 = (za + beta * s1 + gamma) * (zb + beta * s2 + gamma) * zo % r
        e0 = D["e0"]
        const = (PIz - alpha * f2 * (zc + gamma) - alpha * alpha * L1z) % r
        r_c = alg.lincomb(n + 6, [
            (za * zb, C["qM"]), (za, C["qL"]), (zb, C["qR"]), (zc, C["qO"]), (1, C["qC"]), (const, e0),
            (alpha * f1 + alpha * alpha * L1z, z_c), (-alpha * f2 * beta, C["S_sigma3"]),
            (-(zn - 1), t_lo), (-(zn - 1) * zn, t_mid), (-(zn - 1) * zn * zn, t_hi)])
        assert alg.eval(r_c, zeta) == 0, "r(zeta) should be zero"                       # plonk/prover.py:171
        if trace is not None:
            trace.update(beta=beta, gamma=gamma, alpha=alpha, zeta=zeta, v=v, evaluations=dict(ev),
                         a=a_c, b=b_c, c=c_c, z=z_c, PI=PI_c, t=t_c[:3 * n + 6], t_lo=t_lo, t_mid=t_mid,
                         t_hi=t_hi, r=r_c)
        polys = [r_c, a_c, b_c, c_c, C["S_sigma1"], C["S_sigma2"]]
        stride = n + 6
        pack = torch.stack([alg.padded(p, stride) for p in polys]).contiguous()
        zw_ = _native.int_to_words
        # the two openings (plonk/prover.py:184-185) are independent: both go through the pipelined entry point
        # and their witness MSMs overlap; results arrive at the flush
        L = alg.ctx.fp_limbs
        o1 = (np.zeros(2 * L, dtype=np.uint64), np.zeros(1, dtype=np.uint8), np.zeros(4, dtype=np.uint64))
        o2 = (np.zeros(2 * L, dtype=np.uint64), np.zeros(1, dtype=np.uint8), np.zeros(4, dtype=np.uint64))
        mine = [0, 1] if sh is None else sh.mine(2)                          # with several ranks: one opening each
        if 0 in mine:
            alg.ctx.open_device_async(ck.srs, pack.data_ptr(), [p.shape[0] for p in polys], stride, zw_(zeta), zw_(v), *o1)
        zpack = alg.padded(z_c, stride)
        if 1 in mine:
            alg.ctx.open_device_async(ck.srs, zpack.data_ptr(), [z_c.shape[0]], stride, zw_(zeta * int(g) % r), zw_(v), *o2)
        alg.ctx.commit_flush()
        local = {i: kzg._points(o[0], o[1])[0] for i, o in ((0, o1), (1, o2)) if i in mine}
        W_z, W_zw = (local[0], local[1]) if sh is None else sh.gather_points(local, 2)
        return {"commitments": dict(zip(("a", "b", "c"), wire_comms), z=z_comm,
                                    t_lo=t_comms[0], t_mid=t_comms[1], t_hi=t_comms[2]),
                "evaluations": evF,
                "kzg_proofs": {"W_z": W_z, "W_zw": W_zw}}
