"""Multi-GPU use of the engine: one process per GPU (torch.distributed; backend "nccl"
is RCCL on ROCm, "gloo" in the CPU tests).  The reference has nothing distributed;
this is the MI355X design of SURVEY.md section 8e:

  * batch mode  -- the k polynomials of one commit call (3+1+3 per PLONK proof,
    plonk/prover.py:89,113,136) are dealt round-robin to the ranks, each rank runs
    whole MSMs against a replicated SRS, and the k result points (<= 97 bytes each)
    are all-gathered.  No data-path collective.
  * range mode  -- ONE polynomial and the SRS are partitioned by contiguous
    coefficient range; every rank runs a full local Pippenger over its slice and the
    G partial points are all-gathered and added on the host (the EC group law is not
    an RCCL reduction operator): G-1 point additions, latency only.

The local commit is injected (`commit_fn`) so the exchange logic is testable on CPU
with gloo; in production it is KZG.commit on this rank's GPU."""
import torch.distributed as dist


def round_robin(n_items, world):
    """owner rank of each item."""
    return [i % world for i in range(n_items)]


def range_of(rank, world, n):
    """[lo, hi) coefficient range of `rank` when n coefficients are split in `world` contiguous blocks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DistributedCommitter:
    def __init__(self, commit_fn, add_fn, zero_point, group=None):
        """commit_fn(list_of_coefficient_lists) -> list of points (this rank's device);
        add_fn(p, q) -> point (host group law, e.g. KZG.add); zero_point: Z1."""
        self.commit_fn = commit_fn
        self.add_fn = add_fn
        self.zero = zero_point
        self.group = group

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def _all_gather(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def commit_batch(self, polynomials):
        """Every rank passes the same list; returns the full ordered list of commitments on every rank."""
        owners = round_robin(len(polynomials), self.world)
        mine = [i for i, o in enumerate(owners) if o == self.rank]
        local = self.commit_fn([polynomials[i] for i in mine]) if mine else []
        gathered = self._all_gather(list(zip(mine, local)))
        out = [None] * len(polynomials)
        for part in gathered:
            for i, pt in part:
                out[i] = tuple(pt)
        return out

    def open_range(self, begin_fn, finish_fn, z, modulus, n_total):
        """KZG.open of polynomials partitioned by coefficient range (kzg.py:122-159, sharded).

        begin_fn() -> int H_g: value at z of this rank's combined slice polynomial, local indexing
                      (kzg_open_shard_begin);
        finish_fn(carry, first_rank) -> (partial proof point, P(z) or None)   (kzg_open_shard_finish).
        One exchange of a field element per rank, one of a point per rank.  Returns (proof, P(z))."""
        world, rank = self.world, self.rank
        H = self._all_gather(int(begin_fn()))
        lo_hi = [range_of(g, world, n_total) for g in range(world)]
        hi = lo_hi[rank][1]
        carry = sum(H[g] * pow(z, lo_hi[g][0] - hi, modulus) for g in range(rank + 1, world)) % modulus
        part, ev = finish_fn(carry, rank == 0)
        acc = self.zero
        got = self._all_gather((tuple(part), ev))
        for pt, _ in got:
            acc = self.add_fn(acc, tuple(pt))
        return acc, got[0][1]

    def commit_range(self, local_coeffs):
        """local_coeffs: this rank's contiguous slice of ONE polynomial (its SRS shard is what
        commit_fn commits against).  Returns the commitment to the whole polynomial on every rank."""
        part = self.commit_fn([local_coeffs])[0]
        acc = self.zero
        for pt in self._all_gather(tuple(part)):
            acc = self.add_fn(acc, tuple(pt))
        return acc


class DistributedNTT:
    """Four-step NTT / INTT of n = 2^log_n = N1*N2 elements sharded over G ranks by contiguous
    index range (natural order in, natural order out), G | N1 and G | N2:

        all-to-all   rows of the N1 x N2 view  ->  whole columns per rank
        local        column transforms (length N1) + twist w^(t*v)        ops.columns(M, col_base)
        all-to-all   back to whole rows per rank
        local        row transforms (length N2) + final scale             ops.rows(T)
        all-to-all   row t, index b  ->  position b*N1 + t of the natural-order result

    Each all-to-all moves 1/G of the shard to every peer -- one xGMI link per peer, all links
    driven at once (it is per-link bound, so no ring).  The local halves are injected: the GPU
    passes (GpuNttOps -> kzg_ntt_columns_device / kzg_ntt_rows_device) in production, the oracle in
    the gloo tests.  Tensors are int64 [.., 4] views of canonical Fr elements."""

    def __init__(self, ops, group=None, exchange=None):
        self.ops = ops
        self.group = group
        self._exchange = exchange

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def exchange(self, send):
        """send[h] goes to rank h; returns recv with recv[h] = what rank h sent to us."""
        if self._exchange is not None:
            return self._exchange(send)
        if self.world == 1:
            return send
        import torch
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send.contiguous(), group=self.group)
        return recv

    def transform(self, x_local, log_n, world=None, rank=None):
        G = self.world if world is None else world
        g = self.rank if rank is None else rank
        k1 = (log_n + 1) // 2
        k2 = log_n - k1
        N1, N2 = 1 << k1, 1 << k2
        if N1 % G or N2 % G:
            raise ValueError("world size must divide both N1 and N2")
        R1, W = N1 // G, N2 // G
        assert x_local.shape == (R1 * N2, 4)
        # rows -> columns
        send = x_local.view(R1, G, W, 4).permute(1, 0, 2, 3).contiguous()
        M = self.exchange(send).reshape(N1, W, 4)
        self.ops.columns(M, g * W)
        # columns -> rows
        recv = self.exchange(M.view(G, R1, W, 4))
        T = recv.permute(1, 0, 2, 3).contiguous().view(R1, N2, 4)
        self.ops.rows(T)
        # (t, b) -> natural index b*N1 + t
        send = T.view(R1, G, W, 4).permute(1, 0, 2, 3).contiguous()
        recv = self.exchange(send)
        return recv.permute(2, 0, 1, 3).contiguous().view(W * N1, 4)


class GpuNttOps:
    """Local halves of DistributedNTT on this rank's GPU (tensors on the context's device; the
    context must share the torch stream: ctx.set_stream(torch.cuda.current_stream().cuda_stream))."""

    def __init__(self, ctx, log_n, w_words, inverse):
        self.ctx, self.log_n, self.w, self.inverse = ctx, log_n, w_words, inverse

    def columns(self, M, col_base):
        self.ctx.ntt_columns_device(M.data_ptr(), self.log_n, self.w, self.inverse, M.shape[1], col_base)

    def rows(self, T):
        self.ctx.ntt_rows_device(T.data_ptr(), self.log_n, self.w, self.inverse, T.shape[0])
