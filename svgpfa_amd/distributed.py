"""Trial sharding over the GPUs of one node.

Every per-trial quantity is independent across trials, so each rank owns a contiguous block of trials
(balanced by spike count, svgpfa_amd.utils.synthetic.shard_trials) and runs its own engine.  The only
exchange per evaluation is ONE all-reduce (sum) of the engine's `shared` vector
[ELBO, ELL, KL, dC, dd, dtheta] -- 2 218 doubles at the north-star config, latency bound.
Gradients of per-trial parameters (means, cholVecs, Z) never leave their owner.

Backend: "nccl" (= RCCL over xGMI on ROCm) for device tensors, "gloo" in the CPU tests."""
import torch.distributed as dist


def allreduce_shared(shared, group=None):
    """In-place sum of the shared vector over ranks; no-op without an initialised process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if shared.is_cuda and dist.get_backend(group) == "gloo":     # CPU rehearsal of the exchange (tests)
            host = shared.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            shared.copy_(host)
        else:
            dist.all_reduce(shared, op=dist.ReduceOp.SUM, group=group)
    return shared


def allreduce_shared_async(shared, group=None):
    """The same sum, enqueued without blocking the caller's stream: returns a work handle whose wait() makes the current
    stream wait for the result (None when there is nothing to wait for: one rank, or the gloo rehearsal, which is
    synchronous).  For callers that can use the result later than the next launch -- a loop of independent evaluations into
    alternating output buffers -- the exchange then overlaps with the following evaluation's kernels."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if shared.is_cuda and dist.get_backend(group) == "gloo":
            allreduce_shared(shared, group)
            return None
        return dist.all_reduce(shared, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def agree_max(flag, group=None):
    """Maximum of a small non-negative int over ranks (every rank gets it): used to agree on a rank-local failure
    (a non positive-definite Kzz of ONE rank's trials) so that all ranks raise together instead of one rank leaving
    the EM step while the others wait in the next collective."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        import torch
        dev = "cpu" if dist.get_backend(group) == "gloo" else torch.device("cuda", torch.cuda.current_device())
        t = torch.tensor([int(flag)], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return int(t.item())
    return int(flag)


def attach(model, group=None):
    """Make an svgpfa_amd.stats SVLowerBound (holding this rank's trials) reduce over `group`: the shared vector is
    summed once per evaluation, and a failed Cholesky is agreed on once per kernel rebuild."""
    model._reduce = lambda t: allreduce_shared(t, group)
    model._agree = lambda flag: agree_max(flag, group)
    return model
