"""Worker of tests/test_gpu_dist_update.py: one rank of a multi-rank discriminator update on the ONE visible GPU (gloo
rendezvous, device tensors staged through the host by the rehearsal path of distributed.UpdateExchange).  Started as a child
process with RANK / WORLD_SIZE / MASTER_* in the environment; writes what the parent compares into <out_dir>/rank<r>.pt."""

import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

C, BS, EPOCHS, MBS, UPDATES = 166, 512, 2, 2, 2


def make_inputs(rank):
    """Per-rank data: different rollouts, different replay history, different motion datasets (sizes differ too)."""
    g = torch.Generator().manual_seed(100 + rank)
    expert = torch.randn(1500 + 300 * rank, C, generator=g)
    rollouts = [torch.randn(8, 96, C, generator=g) for _ in range(UPDATES)]
    return expert, rollouts


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer, AmpDiscriminatorUpdate, AmpReplayBuffer
    from oracle import disc as odisc

    w = odisc.make_weights(C, seed=4)                      # the same initial replica on every rank
    disc = AmpDiscriminator([(a.cuda(), b.cuda()) for a, b in w], "cuda:0")
    trainer = AmpDiscriminatorTrainer(disc, batch_size=BS, defer_refresh=True)
    replay, motion = AmpReplayBuffer(4000, C, "cuda:0", seed=5), AmpReplayBuffer(2500, C, "cuda:0", seed=6)
    expert, rollouts = make_inputs(rank)
    motion.add_samples(expert.cuda())
    upd = AmpDiscriminatorUpdate(trainer, replay, motion, learning_epochs=EPOCHS, mini_batches=MBS, seed=9 + rank, record_batches=True,
                                 group=dist.group.WORLD)
    batches, losses = [], []
    for r in rollouts:
        losses += [l.clone() for l in upd.update(r.cuda())]
        batches += [torch.stack(b) for b in upd.batches]
    torch.cuda.synchronize()
    mean, var, count = trainer.scaler_state()
    m, v, step = trainer.adam_state()
    x = torch.randn(256, C, generator=torch.Generator().manual_seed(3)).cuda()
    score = disc.style_reward(x, want_logits=True)         # the refreshed inference planes of the trained replica
    torch.save({"weights": [t.cpu() for pair in trainer.weights() for t in pair], "exp_avg": m.cpu(), "exp_avg_sq": v.cpu(), "step": step,
                "mean": mean.cpu(), "var": var.cpu(), "count": count, "losses": torch.stack(losses).cpu(),
                "batches": torch.stack(batches).cpu(), "replay_len": len(replay), "replay_head": replay.memory_index,
                "logits": score["logits"].cpu(), "rows_per_rank": upd.exchange.rows_per_rank,
                "replay_rows": replay.sample(2048)[:].cpu()},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
