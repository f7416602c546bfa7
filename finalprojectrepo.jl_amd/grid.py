"""Implicit global grid for the domain-decomposed 3D diffusion path.

Role of ImplicitGlobalGrid.jl in the reference (init_global_grid / nx_g / x_g / update_halo! /
finalize_global_grid; call sites scripts-part1/part1_kernel_programming.jl:100-101,117,182,187) and of
MPI.Allreduce! in dist_norm_L2 (part1_utils.jl:38), rebuilt for one process per GPU.

Two transports behind the same choreography:
  "rccl"  the product path: the halo exchange and the all-reduce run INSIDE libfpr_hip.so (fpr_comm_init,
          fpr_grid_init, fpr_halo_exchange3d[_begin/_end], fpr_allreduce_sum_dev -- RCCL over xGMI; csrc/comm.hip),
          exactly what the Julia shim calls.  torch.distributed is used for ONE thing: broadcasting the 128-byte
          RCCL unique id at start-up (any backend; bench.py uses gloo as its control plane).
  "dist"  point-to-point through a torch.distributed-like object with the product's pack / unpack callbacks: gloo
          on CPU (tests/test_grid_cpu.py, oracle kernels per shard) and an in-process fake for several emulated
          ranks on one GPU (tests/test_gpu_halo.py).  Same faces, same planes, same order as csrc/comm.hip.

Local arrays are nx*ny*nz including a 1-cell halo on every side that has a neighbour (overlap 2), so
the global grid has dims*(n-2)+2 cells per dimension.  Per pseudo-iteration (GlobalGrid.step):

    compute stream : boundary slabs -> pack x/y faces ----> interior box ----------> unpack halos
    comm stream    :                      wait(compute) -> isend/irecv (RCCL) -> done

i.e. the halo exchange of the freshly written planes overlaps the interior update -- the role of
`@hide_communication (8,8,8)` (:185-188).  Deviation from the reference, documented in DESIGN.md: the
reference calls update_halo!(Hτ) on the OLD buffer (:182,187 before the swap :190); here the NEW buffer
(Hτ2) is exchanged, which makes an N-shard run equal to the single-domain run on the same global grid.
"""
import math


def dims_create(nprocs, ndims=3):
    """MPI.Dims_create-like balanced factorisation, non-increasing: 2->(2,1,1) 4->(2,2,1) 8->(2,2,2)
    (matches part1_scaling_experiments.jl:35-41)."""
    dims = [1] * ndims
    n = nprocs
    f = 2
    factors = []
    while f * f <= n:
        while n % f == 0:
            factors.append(f)
            n //= f
        f += 1
    if n > 1:
        factors.append(n)
    for p in sorted(factors, reverse=True):
        dims[dims.index(min(dims))] *= p
    return tuple(sorted(dims, reverse=True))


ALLFACES = 63          # bit 2*dim + side
ZFACES = 3 << 4


class HaloExchanger:
    """update_halo!(A): exchange one plane per face with each Cartesian neighbour.

    For face (dim, side) the plane at local index 1 (side 0) / n-2 (side 1) is sent to the neighbour,
    whose plane arrives in our halo plane 0 / n-1.  pack(A, face, buf) / unpack(A, face, buf) move a
    plane between the field and a contiguous buffer; z-planes are contiguous in the column-major
    layout and are sent / received in place (zero copy)."""

    def __init__(self, shape, neighbors, rank_of, pack, unpack, new_buffer, dist=None, group=None):
        self.shape = tuple(shape)
        self.neighbors = neighbors  # {face: neighbour coords}
        self.rank_of = rank_of
        self.pack, self.unpack = pack, unpack
        self.dist, self.group = dist, group
        nx, ny, nz = self.shape
        plane = {0: ny * nz, 1: nx * nz, 2: nx * ny}
        self.sendbuf, self.recvbuf = {}, {}
        for face in neighbors:
            d = face >> 1
            if d != 2:
                self.sendbuf[face] = new_buffer(plane[d])
                self.recvbuf[face] = new_buffer(plane[d])

    def faces(self):
        return sorted(self.neighbors)

    @staticmethod
    def _zplane(A, k):
        # A has Julia shape (nx,ny,nz) with strides (1,nx,nx*ny): permute(2,1,0) is C-contiguous
        return A.permute(2, 1, 0)[k]

    def pack_all(self, A, mask=63, stream_sel=0):
        for face in self.faces():
            if (face >> 1) != 2 and (mask >> face) & 1:
                self.pack(A, face, self.sendbuf[face], stream_sel) if stream_sel else self.pack(A, face, self.sendbuf[face])

    def post(self, A, mask=63):
        """Post all sends/receives of already packed planes (z planes in place); returns work handles.
        Order as in csrc/comm.hip post_group: receives low side first, sends high side first, so that with the same
        peer on both sides of a dimension (periodic, dims <= 2) the k-th send pairs with the peer's k-th receive."""
        nz = self.shape[2]
        P2POp = self.dist.P2POp
        recvs, sends = [], []
        for d in range(3):
            for side in (0, 1):
                face = 2 * d + side
                if face in self.neighbors and (mask >> face) & 1:
                    t = self._zplane(A, nz - 1 if side else 0) if d == 2 else self.recvbuf[face]
                    recvs.append(P2POp(self.dist.irecv, t, self.rank_of(self.neighbors[face]), group=self.group))
            for side in (1, 0):
                face = 2 * d + side
                if face in self.neighbors and (mask >> face) & 1:
                    t = self._zplane(A, nz - 2 if side else 1) if d == 2 else self.sendbuf[face]
                    sends.append(P2POp(self.dist.isend, t, self.rank_of(self.neighbors[face]), group=self.group))
        ops = recvs + sends
        return self.dist.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def wait(works):
        for w in works:
            w.wait()

    def unpack_all(self, A, mask=63, stream_sel=0):
        for face in self.faces():
            if (face >> 1) != 2 and (mask >> face) & 1:
                self.unpack(A, face, self.recvbuf[face], stream_sel) if stream_sel else self.unpack(A, face, self.recvbuf[face])

    def update_halo_(self, A):
        """Blocking update_halo!(A)."""
        self.pack_all(A)
        works = self.post(A)
        self.wait(works)
        self.unpack_all(A)


class _DistTransport:
    """Exchange through a torch.distributed-like P2P object (gloo on CPU, in-process fake for emulated ranks)."""

    def __init__(self, gg):
        self.gg = gg

    def begin(self, A, mask=63):
        import torch
        from . import ctx as _ctx

        c = _ctx()
        ex = self.gg.exchanger()
        ex.pack_all(A, mask)                      # compute stream
        c.comm.wait_stream(c.compute)
        with torch.cuda.stream(c.comm):
            return ex.post(A, mask)

    def end(self, A, mask, works):
        import torch
        from . import ctx as _ctx

        c = _ctx()
        ex = self.gg.exchanger()
        with torch.cuda.stream(c.comm):
            ex.wait(works)
        c.compute.wait_stream(c.comm)
        ex.unpack_all(A, mask)

    def prepare(self):
        self.gg.exchanger()       # pack buffers are torch tensors zero-filled on the compute stream

    def comm_post(self, A, mask=63):
        """The exchange as a link of a chain on the COMM stream (fused pairs of a decomposed run): pack and post there."""
        import torch
        from . import ctx as _ctx

        c = _ctx()
        ex = self.gg.exchanger()
        ex.pack_all(A, mask, 1)
        with torch.cuda.stream(c.comm):
            return ex.post(A, mask)

    def comm_complete(self, A, mask, works):
        import torch
        from . import ctx as _ctx

        c = _ctx()
        ex = self.gg.exchanger()
        with torch.cuda.stream(c.comm):
            ex.wait(works)
        ex.unpack_all(A, mask, 1)

    def update_halo_(self, A):
        self.gg.exchanger().update_halo_(A)

    def allreduce_(self, t):
        gg = self.gg
        if gg.dist is not None and gg.nprocs > 1:
            gg.dist.all_reduce(t, op=gg.dist.ReduceOp.SUM, group=gg.group)


class _RcclTransport:
    """The product path: exchange and all-reduce inside libfpr_hip.so (csrc/comm.hip), RCCL over xGMI."""

    def __init__(self, gg):
        from . import ctx as _ctx
        from ._lib import fptr

        self.gg, self.c, self.fptr = gg, _ctx(), fptr
        self.n = (gg.nx, gg.ny, gg.nz)

    def begin(self, A, mask=63):
        self.c.call("fpr_halo_exchange3d_begin", self.fptr(A, 3), *self.n, int(mask))
        return None

    def end(self, A, mask, token):
        self.c.call("fpr_halo_exchange3d_end", self.fptr(A, 3), *self.n, int(mask))

    def prepare(self):
        pass                      # the library owns its pack buffers (fpr_grid_init)

    def comm_post(self, A, mask=63):
        # pack, one RCCL group, unpack -- all in the comm stream's order: complete when the stream gets there
        self.c.call("fpr_halo_exchange3d_comm", self.fptr(A, 3), *self.n, int(mask))
        return None

    def comm_complete(self, A, mask, token):
        pass

    def update_halo_(self, A):
        self.c.call("fpr_halo_exchange3d", self.fptr(A, 3), *self.n)

    def allreduce_(self, t):
        # every RCCL operation of this communicator runs on the comm stream, in program order (the same on all ranks):
        # comm waits for the producers on the compute stream, the compute stream waits for the result
        self.c.call("fpr_stream_wait", 1, 0)
        self.c.call("fpr_allreduce_sum_dev", t.data_ptr(), t.numel(), 1)
        self.c.call("fpr_stream_wait", 0, 1)


def rccl_bootstrap(ctx, rank, world, dist=None, group=None):
    """fpr_comm_init on every rank: rank 0 draws the RCCL unique id, `dist` (torch.distributed, any backend)
    carries its 128 bytes to the others -- the only thing torch.distributed does for the data path."""
    import ctypes as C

    if ctx.L.fpr_comm_size(ctx.h) == world and (world > 1 or ctx.comm_ready):
        return
    buf = C.create_string_buffer(128)
    if rank == 0:
        rc = ctx.L.fpr_comm_get_unique_id(buf)
        if rc != 0:
            raise RuntimeError("fpr_comm_get_unique_id failed (%d)" % rc)
    if world > 1:
        if dist is None:
            raise RuntimeError("several ranks need torch.distributed (any backend) to broadcast the RCCL unique id")
        box = [buf.raw if rank == 0 else None]
        src = 0 if group is None else dist.get_global_rank(group, 0)
        dist.broadcast_object_list(box, src=src, group=group)
        buf = C.create_string_buffer(box[0], 128)
    # this RCCL build prints a five-line version banner on stdout when a communicator is created: send it to stderr, stdout belongs
    # to the caller (bench.py prints ONE JSON line there)
    import os
    import sys

    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        ctx.call("fpr_comm_init", int(rank), int(world), buf)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    ctx.comm_ready = True


_HOSTED_KEEP = []     # ctypes callbacks of hosted transports (must outlive the contexts that hold their addresses)


def hosted_bootstrap(ctx, rank, world, dist, group=None):
    """fpr_comm_init_hosted: the library's exchange logic over a host-staged transport instead of RCCL -- for ranks that SHARE a
    card (RCCL refuses two ranks on one device; a test box has one card).  The bytes travel through `dist` (torch.distributed
    on a CPU backend: gloo): isend of a copy per message (never waits for the peer), a blocking recv per message, per peer in
    posting order.  Same library calls as with RCCL from there on; rates through it are not measurements."""
    import ctypes as C

    import torch

    if ctx.L.fpr_comm_size(ctx.h) == world and ctx.comm_ready:
        return
    pending = []
    own = []          # messages of a rank to itself (a periodic dimension with one rank in it), in posting order

    def grank(peer):
        return peer if group is None else dist.get_global_rank(group, peer)

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
    def send(_user, peer, buf, nbytes):
        try:
            t = torch.frombuffer((C.c_char * nbytes).from_address(buf), dtype=torch.uint8).clone()
            if peer == rank:
                own.append(t)
                return 0
            pending.append((dist.isend(t, grank(peer), group=group), t))
            pending[:] = [(w, x) for w, x in pending if not w.is_completed()]
            return 0
        except Exception:
            return 1

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
    def recv(_user, peer, buf, nbytes):
        try:
            t = torch.frombuffer((C.c_char * nbytes).from_address(buf), dtype=torch.uint8)
            if peer == rank:
                t.copy_(own.pop(0))
                return 0
            dist.recv(t, grank(peer), group=group)
            return 0
        except Exception:
            return 1

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
    def allreduce(_user, x, count):
        try:
            t = torch.frombuffer((C.c_double * count).from_address(C.addressof(x.contents)), dtype=torch.float64)
            dist.all_reduce(t, group=group)
            return 0
        except Exception:
            return 1

    _HOSTED_KEEP.append((send, recv, allreduce, pending))
    ctx.call("fpr_comm_init_hosted", int(rank), int(world), C.cast(send, C.c_void_p), C.cast(recv, C.c_void_p),
             C.cast(allreduce, C.c_void_p), None)
    ctx.comm_ready = True


class GlobalGrid:
    """init_global_grid(nx, ny, nz; dimx, dimy, dimz, periodx, periody, periodz): Cartesian process topology +
    implicit global grid.  transport: "rccl" (library, the product path), "dist" (torch.distributed-like P2P object,
    tests) or None = "rccl" when torch.distributed is initialised with several ranks and a GPU is visible."""

    periods = (0, 0, 0)          # class defaults (a subclass may build the topology by hand)
    transport_kind = "dist"
    _tr = _ex = _sq_host = _sq_shell = _reserve = None
    _pending = False
    _native_pending = False
    _singles = 0

    def __init__(self, nx, ny, nz, dims=None, group=None, use_dist=None, periods=(0, 0, 0), transport=None, drop_faces=0, dist=None):
        """drop_faces (bit 2*dim+side; measurement aid): faces that get no neighbour although the topology has one -- a
        periodic single rank minus its three low faces has the face set of a corner rank of a (2,2,2) decomposition.
        dist: a torch.distributed look-alike to use instead of torch.distributed itself (ThreadWorld.rank_view below)."""
        self.nx, self.ny, self.nz = nx, ny, nz
        if dist is None and (use_dist is None or use_dist):
            try:
                import torch.distributed as _dist

                if _dist.is_available() and _dist.is_initialized():
                    dist = _dist
            except ImportError:
                dist = None
        self.dist, self.group = dist, group
        self.nprocs = dist.get_world_size(group) if dist else 1
        self.me = dist.get_rank(group) if dist else 0
        if dims is None:
            dims = dims_create(self.nprocs)
        dims = tuple(int(d) for d in dims)
        if dims[0] * dims[1] * dims[2] != self.nprocs:
            raise ValueError("dims %s do not match %d processes" % (dims, self.nprocs))
        self.dims = dims
        self.periods = tuple(int(bool(p)) for p in periods)
        self.coords = self.coords_of(self.me)
        self.neighbors = {}
        for d in range(3):
            for side in (0, 1):
                c = list(self.coords)
                c[d] += 1 if side else -1
                if 0 <= c[d] < dims[d]:
                    self.neighbors[2 * d + side] = tuple(c)
                elif self.periods[d]:
                    c[d] %= dims[d]
                    self.neighbors[2 * d + side] = tuple(c)
        for f in range(6):
            if (drop_faces >> f) & 1:
                self.neighbors.pop(f, None)
        self.drop_faces = int(drop_faces)
        if transport is None:
            transport = "dist"
            if self.neighbors and dist is not None:
                try:
                    import torch

                    if torch.cuda.is_available():
                        transport = "rccl"
                except ImportError:
                    pass
        if transport not in ("rccl", "dist", "hosted"):
            raise ValueError("transport must be 'rccl', 'hosted' or 'dist'")
        # "hosted": the library's own exchange logic and choreography (everything "rccl" runs) over a host-staged transport
        # (fpr_comm_init_hosted) -- ranks sharing one card, tests; from here on it IS the "rccl" kind
        self.hosted = transport == "hosted"
        if self.hosted:
            transport = "rccl"
        self.transport_kind = transport
        self._tr = None
        self._ex = None
        self._sq_host = None
        self._sq_shell = None     # norm sums of the shell chain of a fused pair (comm stream)
        self._reserve = None      # override of reserve_cus() (tools, tests)
        self._pending = False     # a fused pair left on the core / comm streams (step2(join=False))
        if transport == "rccl":
            self._init_rccl()

    def _init_rccl(self):
        """fpr_comm_init + fpr_grid_init: the library learns the topology and owns the pack buffers."""
        import ctypes as C
        from . import ctx as _ctx

        c = _ctx()
        if getattr(self, "hosted", False):
            hosted_bootstrap(c, self.me, self.nprocs, self.dist, self.group)
        else:
            rccl_bootstrap(c, self.me, self.nprocs, self.dist, self.group)
        c.set_option("grid_drop_faces", getattr(self, "drop_faces", 0))
        me, npr = C.c_int(0), C.c_int(0)
        dims_o, coords_o = (C.c_int * 3)(), (C.c_int * 3)()
        c.call("fpr_grid_init", self.nx, self.ny, self.nz, *self.dims, *self.periods, C.byref(me), dims_o, C.byref(npr), coords_o)
        ng, nb = (C.c_int * 3)(), (C.c_int * 6)()
        c.call("fpr_grid_info", ng, nb)
        # the library's topology must be the one this object computed (same Cartesian order, same neighbours)
        assert (me.value, npr.value, tuple(dims_o), tuple(coords_o)) == (self.me, self.nprocs, self.dims, self.coords)
        assert tuple(ng) == (self.nx_g(), self.ny_g(), self.nz_g())
        assert {f: nb[f] for f in range(6) if nb[f] >= 0} == {f: self.rank_of(cc) for f, cc in self.neighbors.items()}

    def transport(self):
        if self._tr is None:
            self._tr = _RcclTransport(self) if self.transport_kind == "rccl" else _DistTransport(self)
        return self._tr

    # ---- topology (MPI Cartesian order: last dimension varies fastest) ----
    def coords_of(self, rank):
        d = self.dims
        return (rank // (d[1] * d[2]), (rank // d[2]) % d[1], rank % d[2])

    def rank_of(self, coords):
        d = self.dims
        return (coords[0] * d[1] + coords[1]) * d[2] + coords[2]

    # ---- implicit global grid ----
    def nx_g(self):
        return self.dims[0] * (self.nx - 2) + (0 if self.periods[0] else 2)

    def ny_g(self):
        return self.dims[1] * (self.ny - 2) + (0 if self.periods[1] else 2)

    def nz_g(self):
        return self.dims[2] * (self.nz - 2) + (0 if self.periods[2] else 2)

    def x_g(self, ix, dx, dim=0):
        """Global coordinate of 1-based local index ix (size-n arrays)."""
        n = (self.nx, self.ny, self.nz)[dim]
        return (self.coords[dim] * (n - 2) + (ix - 1)) * dx

    def global_offset(self):
        """0-based global index of local cell (0,0,0)."""
        return tuple(self.coords[d] * ((self.nx, self.ny, self.nz)[d] - 2) for d in range(3))

    # ---- collectives ----
    def allreduce_sum(self, t):
        """MPI.Allreduce!(x, +, comm) of a 1-element tensor (part1_utils.jl:38); returns a float."""
        self.join()
        if self.nprocs > 1:
            self.transport().allreduce_(t)
        return float(t[0].item()) if hasattr(t, "item") else float(t)

    def allreduce_(self, t):
        """In-place sum over all ranks of a small device tensor (several norms in one call); no host sync."""
        self.join()
        if self.nprocs > 1:
            self.transport().allreduce_(t)
        return t

    def barrier(self):
        if self.dist is not None and self.nprocs > 1:
            self.dist.barrier(group=self.group)

    # ---- halo exchange on device arrays (HIP pack/unpack + RCCL) ----
    def exchanger(self):
        if self._ex is None:
            from . import ctx as _ctx
            from ._lib import fptr, fzeros

            c = _ctx()
            nx, ny, nz = self.nx, self.ny, self.nz

            def pack(A, face, buf, stream_sel=0):
                c.call("fpr_halo_pack3d", fptr(A, 3), nx, ny, nz, face, buf.data_ptr(), stream_sel)

            def unpack(A, face, buf, stream_sel=0):
                c.call("fpr_halo_unpack3d", fptr(A, 3), nx, ny, nz, face, buf.data_ptr(), stream_sel)

            self._ex = HaloExchanger((nx, ny, nz), self.neighbors, self.rank_of, pack, unpack,
                                     lambda n: fzeros(n), dist=self.dist, group=self.group)
        return self._ex

    def update_halo_(self, A):
        """update_halo!(A) (part1_kernel_programming.jl:182,187): ordered on the compute stream."""
        self.join()
        if self.neighbors:
            self.transport().update_halo_(A)

    # ---- one pseudo-iteration (kernel + halo exchange + optional norm) ----
    def boundary_boxes(self):
        """Thin boxes (0-based [lo,hi)) holding the interior cells next to faces with a neighbour, and
        the remaining interior box.  Boxes are disjoint and cover the whole interior."""
        n = (self.nx, self.ny, self.nz)
        lo = [1, 1, 1]
        hi = [n[0] - 1, n[1] - 1, n[2] - 1]
        boxes = []
        # peel z faces first (contiguous planes), then y, then x
        for d in (2, 1, 0):
            for side in (0, 1):
                if (2 * d + side) in self.neighbors and hi[d] - lo[d] >= 1:
                    blo, bhi = list(lo), list(hi)
                    if side == 0:
                        bhi[d] = lo[d] + 1
                        lo[d] += 1
                    else:
                        blo[d] = hi[d] - 1
                        hi[d] -= 1
                    boxes.append((tuple(blo), tuple(bhi)))
        return boxes, (tuple(lo), tuple(hi))

    def step(self, Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, norm_scale, sq_dev):
        """Hτ2, dHdτ <- update(Hτ); halos of Hτ2 refreshed; if sq_dev is given it receives the LOCAL
        sum((dHdτ*norm_scale)^2) (all-reduce it with allreduce_sum)."""
        from . import part1

        self.join()
        # a run of single steps (a loop that cannot fuse pairs, a leg of one-iteration launches) undoes the split of the device
        # that fused pairs left behind: unsplit, the interior launch has every compute unit and the exchange runs beside it
        self._singles += 1
        if self._singles == 3 and self.neighbors:
            from . import ctx as _ctx

            _ctx().reserve_comm_cus(0)
        if not self.neighbors:
            if sq_dev is None:
                part1.diffusion_3D_step_τ(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)
            else:
                part1.diffusion_3D_step_τ_norm(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz,
                                               norm_scale, sq_dev)
            return
        st = self.step_begin(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, norm_scale, sq_dev)
        self.step_end(st)

    def can_step2(self, Ht, Hτ, Hτ2, Hout, dHdτ):
        """True if step2 can run two iterations as fused launches on this grid (else call step twice)."""
        from . import part1

        if not part1.can_step_τ2(Ht, Hτ, Hτ2, Hout, dHdτ):
            return False
        # between ranks every decomposed dimension needs room for the one-cell shell next to the halos plus an interior
        n = (self.nx, self.ny, self.nz)
        return all(n[f >> 1] >= 8 for f in self.neighbors)

    def step2(self, Ht, Hτ, Hτ2, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, norm_scale, sq2_dev, join=True):
        """TWO pseudo-iterations: Hout <- update(update(Hτ)), dHdτ <- residual of the second, halos of Hout refreshed.
        Hτ2 plays the reference's second work buffer: only its boundary cells (and, between ranks, its halo planes)
        are used.  Hout must carry Hτ's physical-boundary values.  sq2_dev (2 doubles, or None) receives the LOCAL
        sums of (dHdτ*norm_scale)^2 of the first and second iteration.  Fields bit-identical to two calls of step(); the
        two sums equal step()'s to ~1e-13 relative (another summation order, see include/fpr.h).
        join=False (between ranks): the pair is left on the core / comm streams and the NEXT step2 continues from there
        without passing through the compute stream; call join() (allreduce_ / step / update_halo_ do) before anything
        else reads OR WRITES the fields or the sums (the next pair of a chain reuses the pending pair's x-face strips).
        dHdτ may be None in the one-call form (residual not stored)."""
        from . import part1

        args = (Ht, Hτ, Hτ2, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)
        if not self.neighbors:
            part1.diffusion_3D_step_τ2(*args, norm_scale, sq2_dev)
            return
        if self.transport_kind == "rccl" and self._reserve is None and not self._pending:
            # the product path: the whole choreography in ONE call of the library (fpr_diffusion3d_step2_halo); the phased
            # Python form below is its twin for emulated ranks / the gloo transport and for experiments with the comm share
            from . import ctx as _ctx
            from ._lib import fptr

            self._singles = 0
            _ctx().call("fpr_diffusion3d_step2_halo", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hτ2, 3), fptr(Hout, 3),
                        fptr(dHdτ, 3) if dHdτ is not None else None,   # None: the residual is not stored (its norms still are)
                        self.nx, self.ny, self.nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, float(norm_scale),
                        sq2_dev.data_ptr() if sq2_dev is not None else None, 1 if join else 0)
            self._native_pending = not join
            return
        st = self.step2_begin(*args, norm_scale, sq2_dev)
        self.step2_middle(st)
        self.step2_end(st, join)

    def can_step3(self, Ht, Hτ, Hout, dHdτ):
        """True if step3 can run three iterations as one fused launch (+ the z-shell chain) on this grid: the three-step kernel serves
        the arrays and the rank has neighbours on z-faces only (process grid (1,1,N)) through the library's own transport."""
        from . import part1
        from . import ctx as _ctx
        from ._lib import fptr

        if not self.neighbors:
            return part1.can_step_τ3(Ht, Hτ, Hout, dHdτ)
        if self.transport_kind != "rccl" or self._reserve is not None:
            return False
        c = _ctx()
        return c.L.fpr_diffusion3d_can_step3_halo(c.h, fptr(Ht, 3), fptr(Hτ, 3), fptr(Hout, 3), fptr(dHdτ, 3) if dHdτ is not None else None,
                                                  self.nx, self.ny, self.nz) == 1

    def step3(self, Ht, Hτ, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, norm_scale, sq3_dev, join=True):
        """THREE pseudo-iterations: Hout <- update(update(update(Hτ))), dHdτ <- residual of the third, halos of Hout refreshed.  Hτ and
        Hout are the reference's two ping-pong buffers in either order.  sq3_dev (3 doubles, or None) receives the LOCAL sums of the three
        iterations.  Fields bit-identical to three calls of step().  Between ranks: fpr_diffusion3d_step3_halo (z-slab decompositions;
        join as for step2)."""
        from . import part1

        if not self.neighbors:
            part1.diffusion_3D_step_τ3(Ht, Hτ, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, norm_scale, sq3_dev)
            return
        from . import ctx as _ctx
        from ._lib import fptr

        if self._pending:
            self.join()
        self._singles = 0
        _ctx().call("fpr_diffusion3d_step3_halo", fptr(Ht, 3), fptr(Hτ, 3), fptr(Hout, 3), fptr(dHdτ, 3) if dHdτ is not None else None,
                    self.nx, self.ny, self.nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, float(norm_scale),
                    sq3_dev.data_ptr() if sq3_dev is not None else None, 1 if join else 0)
        self._native_pending = not join

    @property
    def pending(self):
        """True while a fused pair of step2(join=False) sits on the core / comm streams, not yet joined."""
        return self._pending or self._native_pending

    def join(self):
        """Order the compute stream behind a fused pair that step2(join=False) left on the core / comm streams."""
        if self._native_pending:
            from . import ctx as _ctx

            _ctx().call("fpr_diffusion3d_join")
            self._native_pending = False
        if self._pending:
            from . import ctx as _ctx

            # the host waits (a wait parked on the compute stream while pairs are in flight slows the stream waits inside them:
            # fpr_diffusion3d_join); the core stream already waited for the pair's shell chain
            _ctx().synchronize()
            self._pending = False

    # Choreography of two fused iterations between ranks, any Cartesian decomposition (level 1 = the field after the
    # first iteration, never written except where a neighbour needs it).  The SHELL is the one-cell layer of interior
    # cells next to a face with a neighbour (boundary_boxes: disjoint thin boxes), the CORE everything inside it.
    #
    #   core stream :  CORE as ONE fused launch on the compute units the comm stream does not own ------------> join
    #   comm stream :  shell, single steps -> level 1 in Hτ2 -> exchange(Hτ2) -> shell, fused launches (their level-1
    #                  halo cells have just arrived and are read like physical-boundary cells) -> exchange(Hout) --^
    #
    # The device is SPLIT (fpr_reserve_comm_cus): the comm stream owns reserve_cus() compute units, the core stream all the
    # others.  The whole shell chain -- thin launches, pack / unpack kernels, RCCL's send / receive kernels -- runs on its
    # own units beside the core launch and ends long before it: both exchanges of the pair are hidden whatever the links
    # take, and the core is never split.  The core launch serves the plain (tile, chunk) grid with that many workgroups
    # less (fpr_diffusion3d_step2_core).  Round 2 ran the core as two half launches of 255 workgroups on 256 units: an
    # exchange then started when the half beside it drained (tools/cu_share_probe.hip: a second queue's workgroups wait
    # for room in the shader engine they were dealt to).
    # z-faces travel in place, x / y faces through the pack kernels; one-cell-wide x-slabs run in the narrow-box kernel
    # (k_diff3_slab2), y- and z-slabs in the wave-tile kernel -- in this phased form; the one-call form (fpr_diffusion3d_step2_halo)
    # works on the x-shell in compact strips (csrc/diffusion3d_xstrip.hpp).  The phases exist for emulated ranks in one process
    # (tests): every rank posts before any rank completes.
    def reserve_cus(self):
        """Compute units of the comm stream during fused pairs: a multiple of 32 = the same number out of every shader
        engine (a workgroup is dealt to an engine and waits there for room, so a lopsided split puts two workgroups of the
        core launch on one unit: tools/cu_share_probe.hip).  32 carry the shell chain of a rank with z-faces, y-faces or
        one face per dimension in less than the core launch's time; x-faces cost about five z-faces each (lanes along y,
        strided planes), so a rank with more shell work than two of those gets 64 (profiles/r3_step2_faces_overhead.txt: 512^3,
        two faces per periodic dimension: z +9 %, y +11 %, x +27 % at 32 units; xy +38 %, xyz +40 % at 64; one face per
        dimension, a corner rank of (2,2,2): +19 % at 32)."""
        if self._reserve is not None:
            return self._reserve
        work = sum({0: 5, 1: 2, 2: 1}[f >> 1] for f in self.neighbors)
        # no x-faces: 16 (24 for y- and z-faces together) -- shares that are no multiple of 32 use the unmasked core stream, whose core
        # launch lets the workgroups on comm units leave (fpr_reserve_comm_cus); same rule as diff3_comm_units in the library
        if not any(f < 2 for f in self.neighbors):
            return 24 if work > 4 else 16
        return 64 if work > 10 else 32

    def step2_begin(self, Ht, Hτ, Hτ2, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, norm_scale, sq2_dev):
        from . import ctx as _ctx
        from . import part1
        from ._lib import fzeros

        c = _ctx()
        self._singles = 0
        if self._native_pending:      # a pair left pending by the one-call form: the phased form starts from the compute stream
            self.join()
        coef = (dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)
        sqs = None
        if sq2_dev is not None:
            if self._sq_shell is None:
                self._sq_shell = fzeros(2)
            sqs = self._sq_shell
        boxes, core = self.boundary_boxes()
        mask = 0
        for f in self.neighbors:
            mask |= 1 << f
        tr = self.transport()
        fused = (Ht, Hτ, Hτ2, Hout, dHdτ) + coef
        k = self.reserve_cus()
        c.reserve_comm_cus(k)                                 # split the device (no-op once done)
        if self._pending:
            # the previous pair was left on the core / comm streams: its core launch and the addition of its shell sums are
            # in the core stream's order already; the shell chain of this pair has to see that core's output
            c.call("fpr_stream_wait", 1, 2)
        else:
            tr.prepare()                                      # buffers a transport creates (and fills) on the compute stream: before the fork
            c.call("fpr_stream_wait", 1, 0)                   # fork: the pair's inputs are ready
            c.call("fpr_stream_wait", 2, 0)
        # Level 1 on the shell.  The x-slabs (one column wide, lanes along y: every access a cache line of its own) crawl beside
        # a launch that saturates the memory system -- 180 us where they take 25 us alone -- so they run on the CORE stream
        # AHEAD of the core launch; the z- and y-slabs run on the comm stream beside it.  (boundary_boxes peels x last.)
        nxf = sum(1 for f in (0, 1) if f in self.neighbors)
        xboxes = boxes[len(boxes) - nxf:] if nxf else []
        for lo, hi in xboxes:
            part1.diffusion_3D_step_τ_box(Ht, Hτ, Hτ2, dHdτ, *coef, lo, hi, 0.0, None, 2)
        if xboxes:
            c.call("fpr_stream_wait", 1, 2)                   # the comm stream's chain follows them (not the core launch below)
        # the core launch before the thin launches of the comm stream, so that its workgroups are placed first; its sums are written
        part1.diffusion_3D_step_τ2_core(*fused, core[0], core[1], norm_scale, sq2_dev, 2, k, accumulate=False)
        if sqs is not None:
            c.call("fpr_fill_on", sqs.data_ptr(), 0.0, 2, 1)
        for lo, hi in boxes[:len(boxes) - nxf]:               # comm stream: level 1 on the z- and y-slabs
            part1.diffusion_3D_step_τ_box(Ht, Hτ, Hτ2, dHdτ, *coef, lo, hi, 0.0, None, 1)
        works = tr.comm_post(Hτ2, mask)
        return dict(fused=fused, scale=norm_scale, sq=sq2_dev, sqs=sqs, works=works, boxes=boxes, core=core, out=Hout,
                    mid=Hτ2, mask=mask)

    def step2_middle(self, st):
        from . import part1

        tr = self.transport()
        tr.comm_complete(st["mid"], st["mask"], st["works"])
        boxes = list(st["boxes"])
        # the two z-slabs (peeled first, same x / y extent) share one launch
        if 4 in self.neighbors and 5 in self.neighbors:
            (lo0, hi0), (lo1, hi1) = boxes[0], boxes[1]
            part1.diffusion_3D_step_τ2_box(*st["fused"], lo0, hi0, st["scale"], st["sqs"], 1, z2=(lo1[2], hi1[2]))
            boxes = boxes[2:]
        for lo, hi in boxes:
            part1.diffusion_3D_step_τ2_box(*st["fused"], lo, hi, st["scale"], st["sqs"], 1)
        st["works"] = tr.comm_post(st["out"], st["mask"])

    def step2_end(self, st, join=True):
        from . import ctx as _ctx

        c = _ctx()
        self.transport().comm_complete(st["out"], st["mask"], st["works"])
        c.call("fpr_stream_wait", 2, 1)                       # the core stream takes in the shell chain ...
        if st["sq"] is not None:
            c.call("fpr_add_on", st["sq"].data_ptr(), st["sqs"].data_ptr(), 2, 2)    # ... and its sums
        self._pending = True
        if join:
            self.join()

    def step_begin(self, Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, norm_scale, sq_dev):
        """Multi-rank step, first half: boundary slabs, then the exchange of the freshly written planes is posted
        (packs on the compute stream, transfers on the comm stream)."""
        from . import part1

        args = (Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)
        boxes, inner = self.boundary_boxes()
        if sq_dev is not None:
            sq_dev.zero_()
        for lo, hi in boxes:  # 1. boundary slabs first
            part1.diffusion_3D_step_τ_box(*args, lo, hi, norm_scale, sq_dev, 0)
        works = self.transport().begin(Hτ2, ALLFACES)   # 2.
        return (args, inner, norm_scale, sq_dev, works)

    def step_end(self, st):
        """Second half: interior update (overlaps the transfers), join, unpack the received halos."""
        from . import part1

        from . import ctx as _ctx

        args, inner, norm_scale, sq_dev, works = st
        # 3. interior update overlaps the exchange.  On a split device (fused pairs ran before: the comm stream owns a few
        # compute units) it runs on the core stream, whose units the exchange does not compete for
        c = _ctx()
        if c.L.fpr_comm_cus(c.h) > 0:
            c.call("fpr_stream_wait", 2, 0)
            part1.diffusion_3D_step_τ_box(*args, inner[0], inner[1], norm_scale, sq_dev, 2)
            c.call("fpr_stream_wait", 0, 2)
        else:
            part1.diffusion_3D_step_τ_box(*args, inner[0], inner[1], norm_scale, sq_dev, 0)
        # 4. join: the compute stream waits for the transfers, then unpacks the x/y halos
        self.transport().end(args[2], ALLFACES, works)

    def gather_(self, A_host):
        """gather!(A, A_global) onto rank 0 (part1_kernel_programming.jl:223): returns the list of all
        ranks' local arrays (numpy) on rank 0, None elsewhere."""
        if self.nprocs == 1:
            return [A_host]
        if self.dist is None:
            raise RuntimeError("gather_ of host arrays needs torch.distributed; device arrays: gather_global_")
        out = [None] * self.nprocs if self.me == 0 else None
        self.dist.gather_object(A_host, out, dst=0, group=self.group)
        return out


def gather_global_(gg, A):
    """gather!(Array(A), A_global) through the library (fpr_gather3d): rank 0 gets the (nx*dims[0], ny*dims[1],
    nz*dims[2]) host array with every rank's local array, halos included, as a block; the other ranks get None."""
    import ctypes as C

    import numpy as np
    from . import ctx as _ctx
    from ._lib import fptr

    c = _ctx()
    if gg.transport_kind != "rccl":
        c.call("fpr_grid_init", gg.nx, gg.ny, gg.nz, 1, 1, 1, 0, 0, 0, None, None, None, None)
    G = None
    if gg.me == 0:
        G = np.zeros((gg.nx * gg.dims[0], gg.ny * gg.dims[1], gg.nz * gg.dims[2]), order="F")
    c.call("fpr_gather3d", fptr(A, 3), gg.nx, gg.ny, gg.nz, G.ctypes.data_as(C.c_void_p) if G is not None else None)
    return G


def assemble_global(parts, dims):
    """gather!(A, A_global) layout (part1_kernel_programming.jl:144,223): the local arrays, halos included, side by
    side in Cartesian order -> array of shape (nx*dims[0], ny*dims[1], nz*dims[2])."""
    import numpy as np

    nx, ny, nz = parts[0].shape
    G = np.zeros((nx * dims[0], ny * dims[1], nz * dims[2]), order="F")
    for r, a in enumerate(parts):
        c = (r // (dims[1] * dims[2]), (r // dims[2]) % dims[1], r % dims[2])
        G[c[0] * nx:(c[0] + 1) * nx, c[1] * ny:(c[1] + 1) * ny, c[2] * nz:(c[2] + 1) * nz] = a
    return G


def finalize_global_grid():
    """finalize_global_grid(): releases the library's communicator and pack buffers (fpr_comm_finalize);
    torch.distributed, if used for the bootstrap, is owned by the caller."""
    from . import _default_ctx, _tls

    c = getattr(_tls, "ctx", None) or _default_ctx
    if c is not None and getattr(c, "comm_ready", False):
        c.call("fpr_comm_finalize")
        c.comm_ready = False
    return None


class ThreadWorld:
    """N ranks as N THREADS of one process: the part of torch.distributed that hosted_bootstrap and GlobalGrid use (isend / recv /
    all_reduce / barrier / rank / size), over in-process queues.  For rehearsing process grids with more ranks than a one-card box
    admits processes (six): every rank is a thread with its own library context (bind_context) running the library's exchange code and
    pair choreography over fpr_comm_init_hosted.  Not a transport of the product."""

    class _Done:
        @staticmethod
        def is_completed():
            return True

        @staticmethod
        def wait():
            return None

    class ReduceOp:
        SUM, MAX = "sum", "max"

    def __init__(self, world):
        import queue
        import threading

        self.world = world
        self.q = {(s, d): queue.Queue() for s in range(world) for d in range(world)}
        self.bar = threading.Barrier(world)
        self.slots = [None] * world

    def rank_view(self, rank):
        tw = self

        class View:
            ReduceOp = ThreadWorld.ReduceOp

            @staticmethod
            def get_world_size(group=None):
                return tw.world

            @staticmethod
            def get_rank(group=None):
                return rank

            @staticmethod
            def isend(t, peer, group=None):
                tw.q[(rank, peer)].put(t)            # (hosted_bootstrap hands a private copy)
                return ThreadWorld._Done

            @staticmethod
            def recv(t, peer, group=None):
                t.copy_(tw.q[(peer, rank)].get(timeout=120.0))

            @staticmethod
            def all_reduce(t, op="sum", group=None):
                tw.slots[rank] = t.clone()
                tw.bar.wait(timeout=120.0)
                acc = tw.slots[0].clone()
                for k in range(1, tw.world):         # the same order on every rank: identical results everywhere
                    acc = torch_max(acc, tw.slots[k]) if op == "max" else acc + tw.slots[k]
                tw.bar.wait(timeout=120.0)
                t.copy_(acc)

            @staticmethod
            def barrier(group=None):
                tw.bar.wait(timeout=120.0)

        def torch_max(a, b):
            import torch

            return torch.maximum(a, b)

        return View
