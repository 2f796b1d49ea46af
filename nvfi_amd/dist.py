"""Ray-sharded data parallelism: one process per GPU, ONE all-reduce per step.

Every rank holds a full replica of the field, renders its own rays / collocation points, and the only
exchange is a sum of the flat gradient buffer (RCCL over xGMI through torch.distributed backend "nccl";
"gloo" on CPU for the tests) - issued in two pieces so that the large one (plane + render-MLP gradients, final once
the renders are differentiated) travels while the PDE term is still being computed.  The loss means are formed so that the averaged gradient equals the
single-process gradient of the global batch: the render MSE is a mean over equal-sized shards, and the
PDE term - whose per-rank kept count differs - is re-weighted by W * n_kept_r / sum_r n_kept_r.
"""
import torch
import torch.distributed as dist


class RcclComm:
    """RCCL communicator behind the C ABI (nvfi_comm_* / nvfi_allreduce_grads, include/nvfi_hip.h): the data-path collective without
    torch.distributed in the loop.  Bootstrap only needs a way to ship 128 bytes from rank 0 to the others; here that is
    torch.distributed's object broadcast when a process group exists (any backend), or a single process.

    bench.py keeps torch.distributed's all_reduce (the same RCCL underneath) as its default exchange - the launcher contract hands the
    rendezvous to torch.distributed anyway, and that path is the one exercised on the 8-GPU node; `NVFI_ALLREDUCE=abi` switches the
    GradBucket to this communicator."""

    def __init__(self, world=None, rank=None):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        L = _lib.lib()
        have_pg = dist.is_available() and dist.is_initialized()
        self.world = world if world is not None else (dist.get_world_size() if have_pg else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if have_pg else 0)
        if self.world > 1 and not have_pg:
            raise _lib.NvfiError("RcclComm(world>1) ships the 128-byte RCCL id from rank 0 through torch.distributed: call "
                                 "dist.init_process_group (any backend) first, or distribute nvfi_comm_unique_id's bytes yourself")
        ident = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(L.nvfi_comm_unique_id(ident))
        if self.world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_char * 128).from_buffer_copy(box[0])
        self.handle = C.c_void_p()
        _lib.check(L.nvfi_comm_init(C.byref(self.handle), C.c_int(self.world), C.c_int(self.rank), ident))

    def all_reduce_(self, flat, average=True):
        """in place, asynchronous on the current stream"""
        C = self._C
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        self._lib.check(self._lib.lib().nvfi_allreduce_grads(self.handle, self._lib.ptr(flat), C.c_int64(flat.numel()), C.c_int(1 if average else 0),
                                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return flat

    def close(self):
        if self.handle:
            self._lib.lib().nvfi_comm_destroy(self.handle)
            self.handle = None


class GradBucket:
    """One flat fp32 buffer that backs every parameter's .grad (so the step has a single collective)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in self.params:
            if id(p) not in seen:
                seen.add(id(p)); uniq.append(p)
        self.params = uniq
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            chunk = self.flat[off:off + p.numel()]
            if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
                N, C, H, W = p.shape
                g = chunk.view(N, H, W, C).permute(0, 3, 1, 2)   # same strides as the channels_last parameter
            else:
                g = chunk.view(p.shape)
            p.grad = g
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, comm=None):
        if comm is not None:        # RCCL through the C ABI (RcclComm)
            comm.all_reduce_(self.flat, average=True)
            return
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())

    def tail_offset(self, tail_params):
        """Flat offset at which `tail_params` begin, if they are exactly the last parameters of the bucket (else None).
        The head [0, off) can then be reduced early (all_reduce_head_start) while the tail is still being accumulated."""
        ids = {id(p) for p in tail_params}
        off, k = 0, 0
        while k < len(self.params) and id(self.params[k]) not in ids:
            off += self.params[k].numel(); k += 1
        if k == len(self.params) or any(id(p) not in ids for p in self.params[k:]):
            return None
        return off

    def all_reduce_head_start(self, off):
        """Asynchronous sum of flat[:off] (RCCL runs it on its own stream, ordered after the work already queued here); returns
        a handle for all_reduce_finish.  Used to hide the 38 MB plane-gradient exchange behind the PDE kernels."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return None
        return dist.all_reduce(self.flat[:off], op=dist.ReduceOp.SUM, async_op=True)

    def all_reduce_finish(self, handle, off):
        """Sum the tail flat[off:], wait for the head, divide everything by the world size."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        if off < self.flat.numel():
            dist.all_reduce(self.flat[off:], op=dist.ReduceOp.SUM)
        if handle is not None:
            handle.wait()
        self.flat.div_(dist.get_world_size())


def pde_rank_weight(n_kept_local):
    """W * n_r / sum_r n_r  (1.0 for a single process; 0 when nothing is kept anywhere)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1.0
    t = torch.tensor([float(n_kept_local)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    tot = t.clone()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    tot = float(tot.item())
    return 0.0 if tot == 0 else dist.get_world_size() * float(n_kept_local) / tot


def shard_weight(n_local, n_global, world=None):
    """W * n_r / N: the factor that turns a mean over THIS rank's shard (n_r items) into its share of the mean over all N items under the
    averaging all-reduce.  1.0 for equal shards - what bench.py uses (rays // world) - and what a shard_range split of a count that does
    not divide by the world size needs (the first ranks hold one item more)."""
    if world is None:
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return 1.0 if n_global == 0 else world * float(n_local) / float(n_global)


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for this rank (remainder spread over the first ranks)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class PdeGradStage:
    """Small flat buffer that receives the PDE-term gradients of the two velocity nets, so that they can be re-weighted by
    W * n_r / sum_r n_r (computed ON DEVICE from an all-reduced kept count: no host sync) before entering the main bucket."""

    def __init__(self, pde_params):
        self.params = list(pde_params)
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view(p.shape))
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def commit_device(self, pde_out):
        """Same as commit(), with the local kept count taken from the device (pde_out[1] of nvfi_pde_loss): no host value needed."""
        self.commit(pde_out[1:2].detach())

    def commit(self, n_kept_local):
        """p.grad += flat_view * (W * n_r / sum_r n_r)"""
        w = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            n = (n_kept_local.to(torch.float32).reshape(1).clone() if isinstance(n_kept_local, torch.Tensor)
                 else torch.tensor([float(n_kept_local)], dtype=torch.float32, device=self.flat.device))
            tot = n.clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            w = dist.get_world_size() * n / torch.clamp(tot, min=1.0)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            if w is None:
                p.grad.add_(v)
            else:
                p.grad.addcmul_(v, w.expand_as(v))
