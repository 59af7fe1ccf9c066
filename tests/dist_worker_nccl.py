"""Two REAL ranks on RCCL (`torch.distributed` backend nccl), one GPU each: the data-parallel training step of acmil_amd.train.

Launched by tests/test_trainer_gpu.py::test_two_rank_rccl_training_steps (skipped on boxes with fewer than two GPUs) as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/dist_worker_nccl.py
Three steps of  ACMIL_GA.train_step -> GradBucket.allreduce_mean -> FlatAdamW.step  (the reference is single-GPU: its loop is
Step3_WSI_classification_ACMIL.py:189-227; slide-level DP is this repo's addition), step 1 with a bag outside
the split-f16 range on rank 1 only.  Checks, on every rank:
  (i)  the parameters of both ranks are bit-identical after every step (same flat bucket, same optimizer launch);
  (ii) they equal ONE process stepping on the averaged gradients of the same two bags (replayed on this rank's GPU);
  (iii) the flagged step was skipped by BOTH ranks (the flag rides in the bucket) and reported by poll_skipped.
Prints DIST_OK on rank 0.
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acmil_amd import train as T  # noqa: E402


def bag_of(rank, step, bad=False):
    g = torch.Generator().manual_seed(1000 + 10 * step + rank)
    x = torch.randn(400 + 50 * rank + 7 * step, 384, generator=g).half()
    if bad:
        x = x.float()
        x[3, 5] = 1.0e5
    return x, (rank + step) % 3


def setup(dev, seed=7):
    conf = T.Struct(train_epoch=3, warmup_epoch=0, wd=1e-2, lr=1e-3, min_lr=0, n_class=3, n_token=5, n_masked_patch=10, mask_drop=0.6,
                    arch="ga", precision="f16x3", seed=1, D_feat=384, D_inner=128)
    T.set_seed(seed)
    model = T.build_model(conf).to(dev).train()
    bucket = T.GradBucket(list(model.parameters()))
    opt = T.make_optimizer(model, conf, dev, bucket, lr=conf.lr)
    return conf, model, bucket, opt


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    assert world == 2
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    conf, model, bucket, opt = setup(dev, seed=7 + rank)          # different seeds on purpose: the broadcast must make them equal
    T.broadcast_parameters(model, world)
    _, twin, tbucket, topt = setup(dev, seed=99)                  # single-process replica on this rank's GPU
    twin.load_state_dict(model.state_dict())
    uniforms = [torch.rand(5, 10, generator=torch.Generator().manual_seed(50 + s)).to(dev) for s in range(3)]
    for step in range(3):
        bad = (step == 1 and rank == 1)
        x, y = bag_of(rank, step, bad)
        # optimizer handed over as train_one_epoch does: inside a multi-rank job the step must NOT apply the update itself (it would run
        # on this rank's local gradients), whatever in_step says -- it only learns whose fused launches keep its packed weights current
        _, out = model.train_step(x.to(dev).unsqueeze(0), torch.tensor([y], device=dev), uniforms=uniforms[step], guard_flag=opt.guard_flag,
                                  optimizer=opt, track_flag=True, in_step=(step != 2))
        assert out["opt_step_id"] is None, "the step applied the optimizer ahead of the all-reduce"
        bucket.sync_from_grads()
        bucket.allreduce_mean(world)
        sid = opt.step(track_flag=True)
        skipped = opt.poll_skipped(0)
        assert (skipped == [sid]) == (step == 1), (step, skipped)
        # (i) identical parameters on both ranks
        flats = [torch.empty_like(opt.flat) for _ in range(world)]
        dist.all_gather(flats, opt.flat)
        assert torch.equal(flats[0], flats[1]), "ranks diverged at step %d" % step
        # (ii) one process on the averaged gradients (and the averaged flag) of the same two bags
        acc = torch.zeros_like(tbucket.flat)
        for r in range(world):
            xr, yr = bag_of(r, step, step == 1 and r == 1)
            twin.train_step(xr.to(dev).unsqueeze(0), torch.tensor([yr], device=dev), uniforms=uniforms[step], guard_flag=topt.guard_flag)
            tbucket.sync_from_grads()
            acc += tbucket.flat
        tbucket.flat.copy_(acc / world)
        topt.step(track_flag=True)
        topt.poll_skipped(0)
        assert torch.equal(topt.flat, opt.flat), "step %d: DP result != single process on averaged gradients (max diff %g)" % (
            step, (topt.flat - opt.flat).abs().max().item())
    assert opt.skipped_steps == 1 and opt.step_count == 2
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
