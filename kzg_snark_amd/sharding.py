"""Multi-GPU use of the engine: one process per GPU (torch.distributed; backend "nccl"
is RCCL on ROCm, "gloo" in the CPU tests).  The reference has nothing distributed;
this is the MI355X design of SURVEY.md section 8e:

  * batch mode  -- the k polynomials of one commit call (3+1+3 per PLONK proof,
    plonk/prover.py:89,113,136) are dealt round-robin to the ranks, each rank runs
    whole MSMs against a replicated SRS, and the k result points are all-gathered.
    No data-path collective.
  * range mode  -- ONE polynomial and the SRS are partitioned by contiguous
    coefficient range; every rank runs a full local Pippenger over its slice and the
    G partial points are all-gathered and added on the host (the EC group law is not
    an RCCL reduction operator): G-1 point additions, latency only.
  * distributed NTT -- four-step transform with all-to-all transposes (DistributedNTT).

What crosses ranks outside the NTT is a handful of FIXED-SIZE byte records (a G1 point is
97 bytes: x | y as 48-byte little-endian integers and a flag byte; a field element 32 bytes),
all-gathered as uint8 tensors -- on the device for RCCL, on the host for gloo -- never pickled
Python objects.

The local commit is injected (`commit_fn`) so the exchange logic is testable on CPU
with gloo; in production it is KZG.commit on this rank's GPU."""
import torch
import torch.distributed as dist

COORD_BYTES = 48                       # BLS12-381 Fp; BN254 coordinates are zero-padded
POINT_BYTES = 2 * COORD_BYTES + 1      # x | y | flag
FLAG_POINT, FLAG_INFINITY, FLAG_ABSENT = 0, 1, 2
FR_BYTES = 32


def round_robin(n_items, world):
    """owner rank of each item."""
    return [i % world for i in range(n_items)]


def range_of(rank, world, n):
    """[lo, hi) coefficient range of `rank` when n coefficients are split in `world` contiguous blocks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_point(pt):
    """(x, y, z) facade point (normalised: z = 1, or z = 0 for infinity) or None -> 97 bytes."""
    if pt is None:
        return bytes(2 * COORD_BYTES) + bytes([FLAG_ABSENT])
    if int(pt[2]) == 0:
        return bytes(2 * COORD_BYTES) + bytes([FLAG_INFINITY])
    return (int(pt[0]).to_bytes(COORD_BYTES, "little") + int(pt[1]).to_bytes(COORD_BYTES, "little")
            + bytes([FLAG_POINT]))


def unpack_point(rec):
    flag = rec[2 * COORD_BYTES]
    if flag == FLAG_ABSENT:
        return None
    if flag == FLAG_INFINITY:
        return (1, 1, 0)
    return (int.from_bytes(rec[:COORD_BYTES], "little"), int.from_bytes(rec[COORD_BYTES:2 * COORD_BYTES], "little"), 1)


# bench.py --rehearse-collectives: issue every collective even in a one-rank group (the RCCL call path on one GPU)
FORCE_COLLECTIVES = False


_gather_bufs = {}       # (group id, device, payload length, world) -> (send, recv) uint8 tensors, reused across calls


def all_gather_bytes(payload, group=None, always=False):
    """Every rank contributes `payload` (same length everywhere); returns the list of all ranks'
    payloads in rank order.  One all_gather_into_tensor of uint8 records.  Under RCCL ("nccl") the
    records stay on the device between the two copies this needs: ONE host-to-device copy of the
    payload into a buffer kept per (group, length), the collective, ONE device-to-host copy of the
    whole gathered block (round 3 built a tensor per call and fetched every rank's record with its
    own .cpu(): a device synchronisation per rank and exchange).  Under gloo the tensors are host
    memory.  `always`: issue the collective even in a one-rank group (tests: the RCCL call path on
    a single GPU)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not (always or FORCE_COLLECTIVES)):
        return [bytes(payload)]
    world = dist.get_world_size(group)
    n = len(payload)
    if n == 0:
        return [b""] * world
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    key = (id(group) if group is not None else 0, str(dev), n, world)
    bufs = _gather_bufs.get(key)
    if bufs is None:
        if len(_gather_bufs) > 64:
            _gather_bufs.clear()
        send = torch.empty(n, dtype=torch.uint8, device=dev)
        recv = torch.empty(n * world, dtype=torch.uint8, device=dev)
        stage = torch.empty(n, dtype=torch.uint8).pin_memory() if on_gpu else None
        back = torch.empty(n * world, dtype=torch.uint8).pin_memory() if on_gpu else None
        bufs = _gather_bufs[key] = (send, recv, stage, back)
    send, recv, stage, back = bufs
    src = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    if on_gpu:
        stage.copy_(src)
        send.copy_(stage, non_blocking=True)             # ordered on the current stream, like the collective
        dist.all_gather_into_tensor(recv, send, group=group)
        back.copy_(recv, non_blocking=True)
        torch.cuda.current_stream().synchronize()        # the one wait of the exchange
        blob = back.numpy().tobytes()
    else:
        send.copy_(src)
        dist.all_gather_into_tensor(recv, send, group=group)
        blob = recv.numpy().tobytes()
    return [blob[g * n:(g + 1) * n] for g in range(world)]


class DistributedCommitter:
    def __init__(self, commit_fn, add_fn, zero_point, group=None, sum_fn=None):
        """commit_fn(list_of_coefficient_lists) -> list of points (this rank's device);
        add_fn(p, q) -> point (host group law, e.g. KZG.add); zero_point: Z1;
        sum_fn(list of points) -> point, optional: the ranks' partial points added in one call (the library's
        kzg_g1_sum through _native.g1_sum: microseconds per point where KZG.add takes tens)."""
        self.commit_fn = commit_fn
        self.add_fn = add_fn
        self.zero = zero_point
        self.group = group
        self.sum_fn = sum_fn

    def _sum(self, pts):
        if self.sum_fn is not None:
            return self.sum_fn(pts)
        acc = self.zero
        for p in pts:
            acc = self.add_fn(acc, p)
        return acc

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def _gather(self, payload):
        return all_gather_bytes(payload, self.group)

    def commit_batch(self, polynomials):
        """Every rank passes the same list; returns the full ordered list of commitments on every rank."""
        k, world = len(polynomials), self.world
        mine = [i for i in range(k) if i % world == self.rank]
        local = self.commit_fn([polynomials[i] for i in mine]) if mine else []
        per_rank = (k + world - 1) // world                     # fixed record count: absent slots are flagged
        payload = b"".join(pack_point(local[j] if j < len(local) else None) for j in range(per_rank))
        out = [None] * k
        for g, blob in enumerate(self._gather(payload)):
            for j in range(per_rank):
                pt = unpack_point(blob[j * POINT_BYTES:(j + 1) * POINT_BYTES])
                if pt is not None:
                    out[g + j * world] = pt
        return out

    def open_range(self, begin_fn, finish_fn, z, modulus, n_total):
        """KZG.open of polynomials partitioned by coefficient range (kzg.py:122-159, sharded).

        begin_fn() -> int H_g: value at z of this rank's combined slice polynomial, local indexing
                      (kzg_open_shard_begin);
        finish_fn(carry, first_rank) -> (partial proof point, P(z) or None)   (kzg_open_shard_finish).
        One exchange of a field element per rank, one of a point (+ P(z)) per rank.  Returns (proof, P(z))."""
        world, rank = self.world, self.rank
        H = [int.from_bytes(b, "little") for b in self._gather(int(begin_fn()).to_bytes(FR_BYTES, "little"))]
        lo_hi = [range_of(g, world, n_total) for g in range(world)]
        hi = lo_hi[rank][1]
        carry = sum(H[g] * pow(z, lo_hi[g][0] - hi, modulus) for g in range(rank + 1, world)) % modulus
        part, ev = finish_fn(carry, rank == 0)
        payload = pack_point(part) + (b"\x00" * (FR_BYTES + 1) if ev is None
                                      else b"\x01" + int(ev).to_bytes(FR_BYTES, "little"))
        blobs = self._gather(payload)
        ev0 = int.from_bytes(blobs[0][POINT_BYTES + 1:], "little") if blobs[0][POINT_BYTES] == 1 else None
        return self._sum([unpack_point(b[:POINT_BYTES]) for b in blobs]), ev0

    def commit_and_open_range(self, start_commit_fn, collect_commit_fn, begin_fn, finish_fn, z, modulus, n_total):
        """commit_range and open_range of one step with the two local MSMs sharing the commit pipeline:
        start_commit_fn() queues this rank's partial commitment (kzg_commit_device_async) and returns at once, the
        opening's slice evaluation and carry exchange run beside it, finish_fn's MSM follows it through the pipeline,
        and collect_commit_fn() hands back the partial commitment afterwards (kzg_commit_flush).  Two exchanges per
        step instead of three: the field elements, then ONE record per rank holding both partial points (+ P(z)).
        Returns (commitment, (proof, P(z)))."""
        world, rank = self.world, self.rank
        start_commit_fn()
        H = [int.from_bytes(b, "little") for b in self._gather(int(begin_fn()).to_bytes(FR_BYTES, "little"))]
        lo_hi = [range_of(g, world, n_total) for g in range(world)]
        hi = lo_hi[rank][1]
        carry = sum(H[g] * pow(z, lo_hi[g][0] - hi, modulus) for g in range(rank + 1, world)) % modulus
        part_open, ev = finish_fn(carry, rank == 0)
        part_commit = collect_commit_fn()
        payload = pack_point(part_commit) + pack_point(part_open) + (
            b"\x00" * (FR_BYTES + 1) if ev is None else b"\x01" + int(ev).to_bytes(FR_BYTES, "little"))
        blobs = self._gather(payload)
        ev0 = int.from_bytes(blobs[0][2 * POINT_BYTES + 1:], "little") if blobs[0][2 * POINT_BYTES] == 1 else None
        com = self._sum([unpack_point(b[:POINT_BYTES]) for b in blobs])
        prf = self._sum([unpack_point(b[POINT_BYTES:2 * POINT_BYTES]) for b in blobs])
        return com, (prf, ev0)

    def commit_range(self, local_coeffs):
        """local_coeffs: this rank's contiguous slice of ONE polynomial (its SRS shard is what
        commit_fn commits against).  Returns the commitment to the whole polynomial on every rank."""
        part = self.commit_fn([local_coeffs])[0]
        return self._sum([unpack_point(blob) for blob in self._gather(pack_point(part))])


class ProofSharding:
    """BASELINE config 5 on more than one GPU: ONE PLONK proof (plonk/prover.py:24-212) over G ranks.

    Every rank holds the circuit, the witness and a replicated commitment key and walks through the same rounds;
    what a round is made of -- its MSMs -- is dealt:
      * the commitments of a round (plonk/prover.py:89,113,136: 3 + 1 + 3) round-robin over the ranks, item i to
        rank i mod G, each rank running whole MSMs on its own commit pipeline (`commit_batch`);
      * the two opening proofs (plonk/prover.py:184-185), one per rank (`run_dealt`);
      * optionally (`deal_transforms`) the independent transforms of round 1 -- the INTTs of the wire columns and of
        PI and their coset NTTs -- each computed by its owner and broadcast (`dealt_tensors`).  Off by default: a
        2^20-element INTT takes 0.11 ms on one MI355X while its 32 MiB result costs more than that to move over
        one xGMI link (about 0.2 ms at 153 GB/s), so replicated transforms are the faster plan; the MSMs (2.3 ms
        each, 97-byte results) are what pays to deal.
    After every exchange all ranks hold the same points, so they derive the same Fiat-Shamir challenges and end
    with the same proof dict.  What crosses ranks is one all-gather of fixed-size records per round (pack_point),
    plus one broadcast per dealt transform."""

    def __init__(self, group=None, deal_transforms=False, shard_vectors=False):
        """shard_vectors: split the proof's VECTOR work over the ranks as well (plonk_sharded.ShardedProver through
        plonk_sharded.make_prover; power-of-two world sizes) instead of replicating it beside dealt MSMs."""
        self.group = group
        self.deal_transforms = bool(deal_transforms)
        self.shard_vectors = bool(shard_vectors)
        self.exchanges = 0          # collectives issued (tests / bench bookkeeping)

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    @property
    def active(self):
        """more than one rank -- or a one-rank rehearsal that wants every collective issued (FORCE_COLLECTIVES)"""
        return self.world > 1 or FORCE_COLLECTIVES

    def owner(self, i):
        return i % self.world

    def mine(self, k):
        """indices of the k items of a round that this rank computes"""
        return [i for i in range(k) if i % self.world == self.rank]

    def shared_scalars(self, values):
        """Rank 0's list of field elements on every rank: the blinding scalars of a proof (plonk/prover.py:72-75,
        :346) are drawn once, not per rank."""
        payload = b"".join(int(v).to_bytes(FR_BYTES, "little") for v in values)
        self.exchanges += 1
        blob = all_gather_bytes(payload, self.group)[0]
        return [int.from_bytes(blob[i * FR_BYTES:(i + 1) * FR_BYTES], "little") for i in range(len(values))]

    def gather_points(self, local, k):
        """local: {item index: point} for this rank's items of a round of k.  Returns the k points on every rank.
        One all-gather of ceil(k / G) records per rank (absent slots flagged)."""
        world, rank = self.world, self.rank
        per_rank = (k + world - 1) // world
        payload = b"".join(pack_point(local.get(rank + j * world)) for j in range(per_rank))
        self.exchanges += 1
        out = [None] * k
        for g, blob in enumerate(all_gather_bytes(payload, self.group)):
            for j in range(per_rank):
                pt = unpack_point(blob[j * POINT_BYTES:(j + 1) * POINT_BYTES])
                if pt is not None:
                    out[g + j * world] = pt
        assert all(p is not None for p in out), "a rank did not deliver its share of the round"
        return out

    def commit_batch(self, commit_fn, polynomials):
        """commit_fn(list of polynomials) -> list of points on this rank's GPU (KZG.commit against the replicated
        key).  Every rank passes the same list; returns all commitments, in order, on every rank."""
        k = len(polynomials)
        mine = self.mine(k)
        local = commit_fn([polynomials[i] for i in mine]) if mine else []
        return self.gather_points(dict(zip(mine, local)), k)

    def run_dealt(self, thunks):
        """thunks: k callables each returning one point (e.g. the two KZG.open calls); thunk i runs on rank i mod G."""
        mine = self.mine(len(thunks))
        return self.gather_points({i: thunks[i]() for i in mine}, len(thunks))

    def dealt_tensors(self, shapes, compute, like):
        """k independent transforms: compute(i) -> int64 tensor of shape shapes[i] on the owner; every rank returns
        the list of all k results (one broadcast per item from its owner; RCCL moves device tensors, gloo host
        copies).  `like` supplies device and dtype for the receive buffers."""
        out = []
        on_gpu = dist.is_initialized() and dist.get_backend(self.group) == "nccl"
        for i, shape in enumerate(shapes):
            own = self.owner(i) == self.rank
            t = compute(i) if own else torch.empty(shape, dtype=like.dtype, device=like.device)
            if self.active:
                self.exchanges += 1
                src = dist.get_global_rank(self.group, self.owner(i)) if self.group is not None else self.owner(i)
                if t.is_cuda and not on_gpu:         # rehearsal on one GPU: gloo moves host memory only
                    torch.cuda.synchronize(t.device)
                    host = t.cpu()
                    dist.broadcast(host, src=src, group=self.group)
                    if not own:
                        t.copy_(host)
                else:                                # RCCL orders the broadcast after the current torch stream's work
                    dist.broadcast(t, src=src, group=self.group)
            out.append(t)
        return out


class DistributedNTT:
    """Four-step NTT / INTT of n = 2^log_n = N1*N2 elements (N1 = 2^ceil(log_n/2)) over G ranks,
    G | N1 and G | N2.  Input: natural order, rank g holds the contiguous range
    [g*n/G, (g+1)*n/G) = rows g*R1 .. of the N1 x N2 row-major view (R1 = N1/G, W = N2/G).

        pack         [R1][G][W] -> [G][R1][W]                    (the one local copy on the way in)
        all-to-all   rows -> whole columns: every rank now holds [N1][W]
        local        column transforms + twist w^(t*v)           ops.columns(M, col_base)      in place
        all-to-all   columns -> rows; the buffer is sent as it lies ([G][R1][W] blocks are contiguous)
        local        row transforms reading the received blocks where they landed
                                                                  ops.rows_exchange(recv, out, G, blocked)
      layout="transposed" stops here (TWO all-to-alls): the result is [R1][N2], row t_local holding
        result indices b*N1 + (g*R1 + t_local), b = 0..N2-1.  The MSM does not care about order, so a
        commitment of the coefficients uses a key shard generated in the same order
        (kzg_srs_generate_strided: start = g*R1, run_len = N2, inner_stride = N1, outer_stride = 1).
      layout="natural" (fft_ff's ordering, contiguous range per rank) needs what no 2-D split can
        avoid: the row pass writes its output blocked by destination rank, a THIRD all-to-all moves
        the blocks, and one local transpose [N1][W] -> [W][N1] finishes.

    Each all-to-all moves 1/G of the shard to every peer -- one xGMI link per peer, all links
    driven at once (it is per-link bound, so no ring).  The local halves are injected: the GPU
    passes (GpuNttOps) in production, the oracle in the gloo tests.  Tensors are int64 [.., 4]
    views of canonical Fr elements."""

    def __init__(self, ops, group=None, exchange=None):
        self.ops = ops
        self.group = group
        self._exchange = exchange
        self.always_exchange = False     # tests: issue the all-to-all even in a one-rank group (RCCL call path on one GPU)

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def exchange(self, send):
        """send[h] goes to rank h; returns recv with recv[h] = what rank h sent to us.
        `send` must be contiguous: the callers build their buffers in block order."""
        assert send.is_contiguous()
        if self._exchange is not None:
            return self._exchange(send)
        if self.world == 1 and not (self.always_exchange or FORCE_COLLECTIVES):
            return send
        if send.is_cuda and dist.get_backend(self.group) != "nccl":
            # rehearsal on one GPU: gloo moves host memory only
            host = send.cpu()
            got = torch.empty_like(host)
            dist.all_to_all_single(got, host, group=self.group)
            return got.to(send.device)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        return recv

    def transform(self, x_local, log_n, world=None, rank=None, layout="natural"):
        G = self.world if world is None else world
        g = self.rank if rank is None else rank
        k1 = (log_n + 1) // 2
        k2 = log_n - k1
        N1, N2 = 1 << k1, 1 << k2
        if N1 % G or N2 % G:
            raise ValueError("world size must divide both N1 and N2")
        if layout not in ("natural", "transposed"):
            raise ValueError("layout must be 'natural' or 'transposed'")
        R1, W = N1 // G, N2 // G
        assert x_local.shape == (R1 * N2, 4)
        # rows -> columns
        send = x_local.view(R1, G, W, 4).permute(1, 0, 2, 3).contiguous()
        M = self.exchange(send).view(N1, W, 4)
        self.ops.columns(M, g * W)
        # columns -> rows: rows h*R1 .. of M are one contiguous block
        recv = self.exchange(M.view(G, R1, W, 4))
        out = torch.empty_like(recv)
        if layout == "transposed":
            self.ops.rows_exchange(recv, out, G, False)
            return out.view(R1 * N2, 4)
        self.ops.rows_exchange(recv, out, G, True)              # [G][R1][W] over the output index
        got = self.exchange(out)                                # block h: rows of rank h, our W outputs
        # (h, t_local, b_local) -> natural index b_local*N1 + h*R1 + t_local
        return got.view(N1, W, 4).permute(1, 0, 2).contiguous().view(W * N1, 4)


def transposed_index(log_n, world, rank, i):
    """Global coefficient index of element i of the layout="transposed" result on `rank`."""
    k1 = (log_n + 1) // 2
    N1, N2 = 1 << k1, 1 << (log_n - k1)
    R1 = N1 // world
    return (i % N2) * N1 + rank * R1 + i // N2


class GpuNttOps:
    """Local halves of DistributedNTT on this rank's GPU (tensors on the context's device; the
    context must share the torch stream: ctx.bind_torch_stream())."""

    def __init__(self, ctx, log_n, w_words, inverse):
        self.ctx, self.log_n, self.w, self.inverse = ctx, log_n, w_words, inverse

    def columns(self, M, col_base):
        self.ctx.ntt_columns_device(M.data_ptr(), self.log_n, self.w, self.inverse, M.shape[1], col_base)

    def rows_exchange(self, recv, out, world, blocked):
        n_rows = recv.shape[1]
        self.ctx.ntt_rows_exchange_device(recv.data_ptr(), out.data_ptr(), self.log_n, self.w, self.inverse, n_rows,
                                          world, blocked)

    def rows_twist(self, rows, row_base):
        """[n_rows][N2] rows of the transposed layout (global rows row_base ..): row transforms + twist, in place"""
        self.ctx.ntt_rows_twist_device(rows.data_ptr(), self.log_n, self.w, self.inverse, rows.shape[0], row_base)

    def columns_plain(self, M):
        self.ctx.ntt_columns_plain_device(M.data_ptr(), self.log_n, self.w, self.inverse, M.shape[1])


# =====================================================================================================
# Pieces of the VECTOR-sharded prover (kzg_snark_amd/plonk_sharded.py): tensor collectives and the three
# distributed transforms it needs.  Tensors are int64 [m, 4] views of canonical Fr elements; under gloo a
# CUDA tensor makes the round trip through the host (rehearsal on one GPU), under RCCL it stays on the device.
# =====================================================================================================

def _world(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _rank(group=None):
    return dist.get_rank(group) if dist.is_initialized() else 0


def _collective_on(group=None):
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


def all_gather_tensor(t, group=None):
    """[G * m, ...]: every rank's `t` (same shape everywhere) concatenated in rank order."""
    t = t.contiguous()
    if not _collective_on(group):
        return t.clone()
    world = dist.get_world_size(group)
    host_trip = t.is_cuda and dist.get_backend(group) != "nccl"
    src = t.cpu() if host_trip else t
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out.to(t.device) if host_trip else out


def all_to_all_rows(send, send_rows, recv_rows, group=None):
    """Variable all-to-all along dim 0: the first send_rows[0] rows of `send` go to rank 0, the next send_rows[1] to
    rank 1, ..; returns what the ranks sent us, in rank order (recv_rows[h] rows from rank h)."""
    send = send.contiguous()
    if not _collective_on(group):
        assert list(send_rows) == list(recv_rows)
        return send.clone()
    host_trip = send.is_cuda and dist.get_backend(group) != "nccl"
    src = send.cpu() if host_trip else send
    out = torch.empty((sum(recv_rows),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_to_all_single(out, src, output_split_sizes=list(recv_rows), input_split_sizes=list(send_rows), group=group)
    return out.to(send.device) if host_trip else out


class ShardedTransforms:
    """The transforms of a prover whose vectors are split over G ranks (G a power of two):

      natural(x, log_n, w, inverse)        n-point NTT / INTT, contiguous range per rank in AND out
                                           (DistributedNTT, layout "natural": three all-to-alls);
      padded_to_T(x, log_n, log_big, w)    the first n = 2^log_n entries of a zero-padded 2^log_big vector (contiguous
                                           range of n/G per rank) -> its transform in the TRANSPOSED layout of the
                                           2^log_big four-step (two all-to-alls; only the n/N2' non-zero rows of the
                                           N1' x N2' view travel in the first one);
      T_to_natural(x, log_big, w, inverse) transposed layout in -> contiguous range out: the four-step taken the other
                                           way round (row transforms + twist, all-to-all, column transforms,
                                           all-to-all); needs a primitive root w.

    The transposed layout of a 2^L transform over G ranks: rank g holds [R1][N2] (R1 = N1 / G), element (t, b) being
    index b * N1 + g * R1 + t (transposed_index).  Element-wise work does not care about the order as long as every
    vector of one domain uses the same one.

    Below `min_log` (the device passes need more than one 4096-element tile: log_n > 12) or when G does not divide
    the matrix sides, a transform gathers the whole vector, runs it locally and keeps its own part -- the sizes where
    that happens are the sizes where a vector is a few kilobytes.

    make_ops(log_n, w, inverse) -> the local halves for DistributedNTT (GpuNttOps on the device);
    full_ntt(t, w, inverse)     -> whole in-place transform of a local tensor."""

    def __init__(self, make_ops, full_ntt, group=None, min_log=13):
        self.make_ops, self.full_ntt, self.group, self.min_log = make_ops, full_ntt, group, min_log
        self.exchanges = 0

    @property
    def world(self):
        return _world(self.group)

    @property
    def rank(self):
        return _rank(self.group)

    def _distributed(self, log_n):
        G = self.world
        k1 = (log_n + 1) // 2
        return log_n >= self.min_log and (1 << (log_n - k1)) % G == 0 and G & (G - 1) == 0

    def _dntt(self, log_n, w, inverse):
        d = DistributedNTT(self.make_ops(log_n, w, inverse), group=self.group)
        return d

    def t_index(self, log_n, device):
        """global index of every element of this rank's transposed-layout shard of a 2^log_n vector"""
        G, g = self.world, self.rank
        k1 = (log_n + 1) // 2
        N1, N2 = 1 << k1, 1 << (log_n - k1)
        if N1 % G:
            raise ValueError("world size must divide 2^ceil(log_n / 2)")
        i = torch.arange((N1 // G) * N2, device=device)
        return (i % N2) * N1 + g * (N1 // G) + i // N2

    # ---- natural -> natural.  Several vectors of one domain go through ONE set of all-to-alls: their blocks for a peer
    # travel as one message (the local passes run per vector between the exchanges)
    def natural(self, x, log_n, w, inverse):
        return self.natural_batch([x], log_n, w, inverse)[0]

    def _exchange_stacked(self, d, blocks, fresh=False):
        """blocks: B tensors [G, ...] of one shape, block h of each meant for rank h -> what the ranks sent us, per
        vector, [G, ...] contiguous.  One all-to-all for the whole batch.  fresh: the result must not alias the
        caller's tensors (a one-rank exchange hands its input back; the local passes then work in place)."""
        if len(blocks) == 1:
            send = blocks[0].contiguous()
            if fresh and send.data_ptr() == blocks[0].data_ptr():
                send = send.clone()
            return [d.exchange(send)]
        got = d.exchange(torch.stack(blocks, dim=1).contiguous())            # [G][B][...]
        return [got[:, b].contiguous() for b in range(len(blocks))]

    def natural_batch(self, xs, log_n, w, inverse):
        n, G, g = 1 << log_n, self.world, self.rank
        m = n // G
        assert all(x.shape[0] == m for x in xs)
        if not self._distributed(log_n):
            self.exchanges += 1
            full = all_gather_tensor(torch.stack(xs, dim=1).contiguous(), self.group)        # [n][B][4]
            out = []
            for b in range(len(xs)):
                v = full[:, b].contiguous()
                self.full_ntt(v, w, inverse)
                out.append(v[g * m:(g + 1) * m].clone())
            return out
        k1 = (log_n + 1) // 2
        N1, N2 = 1 << k1, 1 << (log_n - k1)
        if N1 % G or N2 % G:
            raise ValueError("world size must divide both sides of the four-step matrix")
        R1, W = N1 // G, N2 // G
        d = self._dntt(log_n, w, inverse)
        self.exchanges += 3
        Ms = [t.view(N1, W, 4) for t in
              self._exchange_stacked(d, [x.view(R1, G, W, 4).permute(1, 0, 2, 3) for x in xs], fresh=True)]   # rows -> columns
        for M in Ms:
            d.ops.columns(M, g * W)
        recvs = self._exchange_stacked(d, [M.view(G, R1, W, 4) for M in Ms])                          # columns -> rows
        outs = []
        for recv in recvs:
            out = torch.empty_like(recv)
            d.ops.rows_exchange(recv, out, G, True)                       # [G][R1][W] over the output index
            outs.append(out)
        gots = self._exchange_stacked(d, outs)                             # block h: rows of rank h, our W outputs
        return [t.view(N1, W, 4).permute(1, 0, 2).contiguous().view(W * N1, 4) for t in gots]

    # ---- first n entries of a zero-padded 2^log_big vector -> transposed layout
    def padded_to_T(self, x, log_n, log_big, w):
        return self.padded_to_T_batch([x], log_n, log_big, w)[0]

    def padded_to_T_batch(self, xs, log_n, log_big, w):
        n, G, g = 1 << log_n, self.world, self.rank
        m = n // G
        assert all(x.shape[0] == m for x in xs) and log_big >= log_n
        k1 = (log_big + 1) // 2
        N1, N2 = 1 << k1, 1 << (log_big - k1)
        if self._distributed(log_big) and m % N2 == 0 and N1 % G == 0:
            R1, W, Rh = N1 // G, N2 // G, m // N2                      # my rows of the N1 x N2 view: g * Rh ..
            d = self._dntt(log_big, w, False)
            gots = self._exchange_stacked(d, [x.view(Rh, G, W, 4).permute(1, 0, 2, 3) for x in xs], fresh=True)
            Ms = []
            for got in gots:                                           # [G][Rh][W]: rows h * Rh + r, my W columns
                M = torch.zeros((N1, W, 4), dtype=got.dtype, device=got.device)
                M[:G * Rh] = got.view(G * Rh, W, 4)
                d.ops.columns(M, g * W)
                Ms.append(M)
            recvs = self._exchange_stacked(d, [M.view(G, R1, W, 4) for M in Ms])
            outs = []
            for recv in recvs:
                out = torch.empty_like(recv)
                d.ops.rows_exchange(recv, out, G, False)
                outs.append(out.view(R1 * N2, 4))
            self.exchanges += 2
            return outs
        self.exchanges += 1
        gathered = all_gather_tensor(torch.stack(xs, dim=1).contiguous(), self.group)             # [n][B][4]
        idx = self.t_index(log_big, xs[0].device)
        outs = []
        for b in range(len(xs)):
            full = torch.zeros((1 << log_big, 4), dtype=xs[0].dtype, device=xs[0].device)
            full[:n] = gathered[:, b]
            self.full_ntt(full, w, False)
            outs.append(full.index_select(0, idx))
        return outs

    # ---- transposed layout -> natural range (then an ordinary distributed transform)
    def T_to_natural(self, x, log_big, w, inverse):
        G, g = self.world, self.rank
        k1 = (log_big + 1) // 2
        N1, N2 = 1 << k1, 1 << (log_big - k1)
        R1, W = N1 // G, N2 // G
        assert x.shape[0] == R1 * N2
        if self._distributed(log_big):
            # the transform taken the other way round: local row transforms + twist, rows -> columns, column
            # transforms, columns -> rows: two all-to-alls (four when the data is first brought into range order)
            d = self._dntt(log_big, w, inverse)
            rows = x.clone().view(R1, N2, 4)
            d.ops.rows_twist(rows, g * R1)
            M = d.exchange(rows.view(R1, G, W, 4).permute(1, 0, 2, 3).contiguous()).view(N1, W, 4)   # my W columns, whole
            d.ops.columns_plain(M)                                        # element (alpha, beta) = X[alpha N2 + beta]
            recv = d.exchange(M.view(G, R1, W, 4))                        # block h: my rows alpha, rank h's columns
            self.exchanges += 2
            return recv.permute(1, 0, 2, 3).contiguous().view(R1 * N2, 4)
        self.exchanges += 1
        parts = all_gather_tensor(x, self.group).view(G, R1 * N2, 4)
        full = torch.empty((1 << log_big, 4), dtype=x.dtype, device=x.device)
        i = torch.arange(R1 * N2, device=x.device)
        for h in range(G):
            full[(i % N2) * N1 + h * R1 + i // N2] = parts[h]
        self.full_ntt(full, w, inverse)
        m = (1 << log_big) // G
        return full[g * m:(g + 1) * m].clone()

    # ---- neighbours in the transposed layout: y[i] = x[(i + shift) mod 2^log_big], 0 < shift <= R1
    def T_shift(self, x, log_big, shift):
        G, g = self.world, self.rank
        k1 = (log_big + 1) // 2
        N1, N2 = 1 << k1, 1 << (log_big - k1)
        R1 = N1 // G
        if not 0 < shift <= R1:
            raise ValueError("shift must be within one rank's rows")
        rows = x.view(R1, N2, 4)
        self.exchanges += 1
        heads = all_gather_tensor(rows[:shift].contiguous(), self.group).view(G, shift, N2, 4)
        nxt = heads[(g + 1) % G]
        if g == G - 1:          # residues wrap: index b * N1 + (rho + shift - N1) + N1 = (b + 1) * N1 + ..
            nxt = torch.roll(nxt, shifts=-1, dims=1)
        return torch.cat([rows[shift:], nxt], dim=0).contiguous().view(R1 * N2, 4)
