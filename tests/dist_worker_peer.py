"""Two ranks, the DIRECT gradient reduction (acmil_amd/peer.py + csrc/peer.hip), able to run on ONE GPU.

Launched by tests/test_trainer_gpu.py::test_two_rank_direct_reduce_on_one_gpu as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/dist_worker_peer.py
Both ranks use cuda:(rank % device_count) -- the same GPU on a 1-GPU box: the peers' slots and flag arrays are mapped through CUDA IPC
exactly as across GPUs, only the xGMI hop is missing.  The control plane (IPC handles, checks) runs on gloo.
Three steps of  ACMIL_GA.train_step -> FlatAdamW.step (publish + wait + rank-ordered reduce + AdamW in the optimizer launch), step 1
with a bag outside the split-f16 range on rank 1 only.  Checks, on every rank: (i) bit-identical parameters on both ranks after every
step; (ii) equal to ONE process stepping on the averaged gradients of the same two bags; (iii) the flagged step skipped by BOTH ranks;
(iv) no torch.distributed collective touched the bucket (allreduce_mean is a no-op with a reducer).  Prints PEER_OK on rank 0.
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acmil_amd import train as T  # noqa: E402
from dist_worker_nccl import bag_of  # noqa: E402


def setup(dev, seed, rank=None, world=1):
    conf = T.Struct(train_epoch=3, warmup_epoch=0, wd=1e-2, lr=1e-3, min_lr=0, n_class=3, n_token=5, n_masked_patch=10, mask_drop=0.6,
                    arch="ga", precision="f16x3", seed=1, D_feat=384, D_inner=128)
    T.set_seed(seed)
    model = T.build_model(conf).to(dev).train()
    bucket = T.GradBucket(list(model.parameters()))
    if rank is not None:
        assert bucket.enable_direct(rank, world), "direct reduction could not be set up"
    opt = T.make_optimizer(model, conf, dev, bucket, lr=conf.lr)
    return conf, model, bucket, opt


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 2
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    conf, model, bucket, opt = setup(dev, 7, rank, world)         # same seed: identical initial parameters (no NCCL broadcast here)
    assert opt.peer is bucket.peer and opt.peer is not None
    _, twin, tbucket, topt = setup(dev, 99)                        # single-process replica on this rank's GPU
    twin.load_state_dict(model.state_dict())
    uniforms = [torch.rand(5, 10, generator=torch.Generator().manual_seed(50 + s)).to(dev) for s in range(3)]
    for step in range(3):
        bad = (step == 1 and rank == 1)
        x, y = bag_of(rank, step, bad)
        # (optimizer handed over as train_one_epoch does: in a multi-rank job the step never applies the update itself)
        _, out = model.train_step(x.to(dev).unsqueeze(0), torch.tensor([y], device=dev), uniforms=uniforms[step], guard_flag=opt.guard_flag,
                                  optimizer=opt, track_flag=True)
        assert out["opt_step_id"] is None, "the step applied the optimizer ahead of the gradient reduction"
        bucket.sync_from_grads()
        before = bucket.flat.clone()
        bucket.allreduce_mean(world)                               # no-op: the optimizer launch reduces
        assert torch.equal(before, bucket.flat)
        sid = opt.step(track_flag=True)
        skipped = opt.poll_skipped(0)
        if opt.peer is None:
            # The first-step check found the direct reduction unequal to all_reduce and fell back.  With both ranks on ONE GPU that
            # is a defect; across real links (a box with several GPUs: the path has never run there) it is the fail-safe doing its
            # job: the ranks must still agree bit for bit, and the run ends here.
            assert torch.cuda.device_count() > 1, "direct reduction failed its first-step check on a single GPU: %s" % bucket.peer.verdict
            flats = [torch.empty_like(opt.flat).cpu() for _ in range(world)]
            dist.all_gather(flats, opt.flat.cpu())
            assert torch.equal(flats[0], flats[1]), "ranks diverged after the fallback"
            bucket.peer.close()
            if rank == 0:
                print("PEER_OK fallback (%s)" % bucket.peer.verdict)
            dist.destroy_process_group()
            return
        opt.peer.check()
        assert (skipped == [sid]) == (step == 1), (step, skipped)
        flats = [torch.empty_like(opt.flat).cpu() for _ in range(world)]
        dist.all_gather(flats, opt.flat.cpu())
        assert torch.equal(flats[0], flats[1]), "ranks diverged at step %d" % step
        acc = torch.zeros_like(tbucket.flat)
        for r in range(world):
            xr, yr = bag_of(r, step, step == 1 and r == 1)
            twin.train_step(xr.to(dev).unsqueeze(0), torch.tensor([yr], device=dev), uniforms=uniforms[step], guard_flag=topt.guard_flag)
            tbucket.sync_from_grads()
            acc += tbucket.flat
        tbucket.flat.copy_(acc / world)
        topt.step(track_flag=True)
        topt.poll_skipped(0)
        assert torch.equal(topt.flat, opt.flat), "step %d: direct DP result != single process on averaged gradients (max diff %g)" % (
            step, (topt.flat - opt.flat).abs().max().item())
    assert opt.skipped_steps == 1 and opt.step_count == 2
    # many steps back to back without host synchronisation in between: the double-buffered slots and the flags must hold up
    for step in range(3, 43):
        x, y = bag_of(rank, step % 5)
        model.train_step(x.to(dev).unsqueeze(0), torch.tensor([y], device=dev), guard_flag=opt.guard_flag)
        bucket.sync_from_grads()
        opt.step()
    torch.cuda.synchronize()
    opt.peer.check()
    flats = [torch.empty_like(opt.flat).cpu() for _ in range(world)]
    dist.all_gather(flats, opt.flat.cpu())
    assert torch.equal(flats[0], flats[1]) and torch.isfinite(flats[0]).all(), "ranks diverged in the unsynchronised run"
    assert opt.peer.verified and opt.peer.verdict.startswith("first step equals all_reduce"), opt.peer.verdict
    memory = opt.peer.memory
    opt.peer.close()                                               # collective teardown: drain, meet, unmap
    # ---- the first-step self-check catching a wrong reduction: rank 1 sees a perturbed bucket (what a stale remote line would look
    # like); BOTH ranks must undo the step, redo it on the collective's result and stay on torch.distributed afterwards
    conf2, model2, bucket2, opt2 = setup(dev, 7, rank, world)
    _, twin2, tbucket2, topt2 = setup(dev, 99)
    twin2.load_state_dict(model2.state_dict())
    opt2.peer._selfcheck_perturb = (rank == 1)
    for step in range(2):
        x, y = bag_of(rank, step)
        model2.train_step(x.to(dev).unsqueeze(0), torch.tensor([y], device=dev), uniforms=uniforms[step], guard_flag=opt2.guard_flag)
        bucket2.sync_from_grads()
        if step > 0:                                               # after the fallback the bucket is reduced by the collective again
            before = bucket2.flat.clone()
            cpu = bucket2.flat.cpu(); dist.all_reduce(cpu); bucket2.flat.copy_((cpu / world).to(dev))      # (gloo control plane: host all-reduce)
            assert not torch.equal(before, bucket2.flat)
        opt2.step(track_flag=True)
        opt2.poll_skipped(0)
        assert opt2.peer is None and bucket2.peer.owner is None and bucket2.peer.verdict.startswith("mismatch"), bucket2.peer.verdict
        acc = torch.zeros_like(tbucket2.flat)
        for r in range(world):
            xr, yr = bag_of(r, step)
            twin2.train_step(xr.to(dev).unsqueeze(0), torch.tensor([yr], device=dev), uniforms=uniforms[step], guard_flag=topt2.guard_flag)
            tbucket2.sync_from_grads()
            acc += tbucket2.flat
        tbucket2.flat.copy_(acc / world)
        topt2.step(track_flag=True)
        topt2.poll_skipped(0)
        assert torch.equal(topt2.flat, opt2.flat), "fallback step %d != single process on averaged gradients" % step
    bucket2.peer.check()                                           # a reducer that fell back has nothing to report (no raise)
    bucket2.peer.close()
    dist.barrier()
    if rank == 0:
        print("PEER_OK memory=%s" % memory)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
