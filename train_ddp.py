#!/usr/bin/env python
"""Multi-GPU entry point: the reference's main.py hard-wires single-process
nn.DataParallel (main.py:79); this is the one-process-per-GPU equivalent with the same
opts.py flags, driven by torchrun:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        train_ddp.py <class_file> RGB <src_list> <tgt_list> <val_list> --frame_aggregation trn-m ... \
        [--synthetic 1438 840]

`-b Bs Bt Bv` are GLOBAL batch sizes (as in the reference); each rank takes a contiguous
shard of every batch, zero-padded to a static per-rank size.  Loss means use global row
counts and gradients are summed by one RCCL all-reduce per step (ta3n_amd/parallel.py).
Schedules follow main.py: beta (main.py:350-352), DANN learning rate (main.py:620-621)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ta3n_amd import parallel  # noqa: E402
from ta3n_amd.engine import TrainEngine, beta_dann, flags_from_options, lr_dann  # noqa: E402
from ta3n_amd.models import ARCH_FEATURE_DIM  # noqa: E402
from ta3n_amd.opts import parser  # noqa: E402


def synthetic_loader(n_videos, batch, T, D, C, seed):
    """Batches of half-normal features [b,T,D] + labels, like TSNDataSet items stacked by a DataLoader."""
    g = torch.Generator().manual_seed(seed)
    n_steps = max(n_videos // batch, 1)
    for _ in range(n_steps):
        yield torch.randn(batch, T, D, generator=g).abs_(), torch.randint(0, C, (batch,), generator=g)


def list_loader(list_file, batch, T, seed):
    from ta3n_amd.dataset import TSNDataSet
    n = sum(1 for _ in open(list_file))
    ds = TSNDataSet("", list_file, num_dataload=n, num_segments=T, new_length=1, modality="RGB", random_shift=False,
                    test_mode=True)
    g = torch.Generator().manual_seed(seed)             # same permutation on every rank: shards are disjoint slices
    sampler = torch.utils.data.RandomSampler(ds, generator=g)
    return torch.utils.data.DataLoader(ds, batch_size=batch, sampler=sampler, num_workers=0, drop_last=False)


def main():
    parser.add_argument("--synthetic", type=int, nargs=2, default=None, metavar=("N_SRC", "N_TGT"),
                        help="train on synthetic features instead of the list files")
    parser.add_argument("--arithmetic", choices=("f32", "bf16"), default="f32",
                        help="f32: the reference's arithmetic (fp32 MFMA); bf16: contraction operands rounded to bf16 and read from bf16 "
                             "twins, fp32 accumulation / parameters / optimiser (BASELINE configs[1])")
    parser.add_argument("--graph", action="store_true", help="replay a captured hipGraph (default: eager launches, faster here)")
    parser.add_argument("--feature_store", type=str, nargs="+", default=None, metavar="PREFIX",
                        help="packed feature stores (ta3n_amd.feature_store.pack): SRC TGT [VAL]; batches are assembled on the GPU")
    args = parser.parse_args()
    rank, local_rank, world = parallel.init_distributed()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    num_class = len([x for x in open(args.class_file)]) if os.path.exists(args.class_file) else 12
    T, D = args.num_segments, ARCH_FEATURE_DIM[args.arch]
    if args.frame_aggregation != "trn-m" or args.baseline_type != "video":
        raise SystemExit("train_ddp.py implements the TA3N hot path: --frame_aggregation trn-m --baseline_type video")
    Bs_g, Bt_g = args.batch_size[0], args.batch_size[1]
    Bs, Bt = parallel.padded_shard_size(Bs_g, world), parallel.padded_shard_size(Bt_g, world)
    flags = flags_from_options(args.place_adv, args.add_loss_DA, args.use_attn, args.adv_DA, args.use_target)
    eng = TrainEngine(Bs, Bt, T, D, args.fc_dim, num_class, flags=flags, dropout_i=args.dropout_i,
                      dropout_v=args.dropout_v, momentum=args.momentum, weight_decay=args.weight_decay,
                      clip=args.clip_gradient, device=dev, bf16=(args.arithmetic == "bf16"), bf16_store=(args.arithmetic == "bf16"))
    from ta3n_amd.models import VideoModel
    torch.manual_seed(1)
    model = VideoModel(num_class, args.baseline_type, args.frame_aggregation, args.modality, train_segments=T,
                       val_segments=T, base_model=args.arch, add_fc=args.add_fc, fc_dim=args.fc_dim,
                       dropout_i=args.dropout_i, dropout_v=args.dropout_v, partial_bn=not args.no_partialbn,
                       use_bn=args.use_bn, ens_DA=args.ens_DA, use_attn=args.use_attn, verbose=False)
    eng.load_state(model.state_dict())                  # reference initialisation under torch.manual_seed(1)
    parallel.broadcast_(eng.P)
    eng.refresh_bf16(params=True)
    if args.synthetic:
        n_src, n_tgt = args.synthetic
    elif args.feature_store:
        n_src, n_tgt = 1, 1                              # set from the stores below
    else:
        n_src, n_tgt = sum(1 for _ in open(args.train_source_list)), sum(1 for _ in open(args.train_target_list))
    steps_per_epoch = max(n_src // Bs_g, 1)
    total_steps = args.epochs * steps_per_epoch
    captured = False
    stores = None
    if args.feature_store:
        from ta3n_amd.feature_store import FeatureStore
        stores = [FeatureStore(pfx, D, dev) for pfx in args.feature_store]
        n_src, n_tgt = len(stores[0]), len(stores[1])
        steps_per_epoch = max(n_src // Bs_g, 1)
        total_steps = args.epochs * steps_per_epoch

    def store_loader(n_videos, batch, seed):
        """Video ids of one epoch: a seeded permutation (RandomSampler, main.py:176-190), identical on every rank."""
        perm = torch.randperm(n_videos, generator=torch.Generator().manual_seed(seed))
        for b in range(max(n_videos // batch, 1)):
            yield perm[b * batch:(b + 1) * batch].to(torch.int32)

    def validate(epoch):
        """main.validate (main.py:669-761) on rank 0: eval-mode forward + device-side CE / top-k / confusion matrix."""
        val = stores[2]
        for b0 in range(0, len(val), Bs):
            ids = torch.arange(b0, min(b0 + Bs, len(val)), dtype=torch.int32, device=dev)
            x, y = val.gather(ids, T)
            eng.evaluate_batch(x, y, reset=(b0 == 0))
        r = eng.eval_results()
        print(f"Test: [{epoch}] Prec@1 {r['prec1']:.3f} Prec@5 {r['prec5']:.3f} Loss {r['loss']:.5f} ({r['n']} videos)", flush=True)

    t_start = time.time()
    for epoch in range(1, args.epochs + 1):
        if stores:
            src = ((ids, None) for ids in store_loader(n_src, Bs_g, 1000 + epoch))
            tgt = ((ids, None) for ids in store_loader(max(n_tgt, 1), Bt_g, 2000 + epoch))
        elif args.synthetic:
            src = synthetic_loader(n_src, Bs_g, T, D, num_class, 1000 + epoch)
            tgt = synthetic_loader(max(n_tgt, Bt_g * steps_per_epoch), Bt_g, T, D, num_class, 2000 + epoch)
        else:
            src = list_loader(args.train_source_list, Bs_g, T, 1000 + epoch)
            tgt = list_loader(args.train_target_list, Bt_g, T, 2000 + epoch)
        for i, ((xs, ys), (xt, _)) in enumerate(zip(src, tgt)):
            p = float(i + epoch * steps_per_epoch) / total_steps                      # main.py:350
            bd = beta_dann(p)
            beta = [bd if b < 0 else b for b in args.beta]                            # main.py:352
            lo, hi = parallel.shard_range(xs.size(0), world, rank)                    # this rank's videos
            lo_t, hi_t = parallel.shard_range(xt.size(0), world, rank)
            if stores:                                                                # batch assembled on the device
                eng.X.zero_()
                if eng.bf16_store:
                    eng.refresh_bf16(x=True)                                            # twin of the zeroed (padding) rows
                stores[0].gather_into(eng, xs[lo:hi].to(dev), 0, labels_out=eng._labels[:Bs])   # features + their bf16 twin
                stores[1].gather_into(eng, xt[lo_t:hi_t].to(dev), Bs)
            else:
                xs_r = torch.zeros(Bs, T, D); xs_r[: hi - lo] = xs[lo:hi]
                xt_r = torch.zeros(Bt, T, D); xt_r[: hi_t - lo_t] = xt[lo_t:hi_t]
                ys_r = torch.zeros(Bs, dtype=torch.long); ys_r[: hi - lo] = ys[lo:hi]
                eng.set_batch(xs_r.to(dev, non_blocking=True), xt_r.to(dev, non_blocking=True), ys_r.to(dev))
            lr = args.lr if (epoch == 1 and i == 0) or args.lr_adaptive != "dann" else eng._lr_next
            if not captured and args.graph:
                eng.set_hyper(beta, args.gamma, lr)
                eng.capture()
                captured = True
            eng.train_step(beta, args.gamma, lr, valid_source=hi - lo, valid_target=hi_t - lo_t,
                           global_source=xs.size(0), global_target=xt.size(0))
            eng._lr_next = lr_dann(args.lr, p) if args.lr_adaptive == "dann" else lr   # main.py:620-621
            if rank == 0 and i % max(args.print_freq, 1) == 0:
                L = eng.losses()          # one host sync every print_freq steps (the reference syncs 5-6x per step)
                print(f"Train: [{epoch}][{i}/{steps_per_epoch}] lr {lr:.5f} loss {L['loss']:.4f} loss_c {L['loss_c']:.4f} "
                      f"loss_a {L['loss_adv_rel'] + L['loss_adv_vid'] + L['loss_adv_frm']:.4f} loss_e {L['loss_e']:.4f} "
                      f"beta {beta[0]:.3f},{beta[1]:.3f},{beta[2]:.3f}", flush=True)
        if stores and len(stores) > 2 and rank == 0:
            validate(epoch)
    torch.cuda.synchronize(dev)
    if rank == 0:
        print(f"total training time: {time.time() - t_start:.1f}s")
        if args.save_model and args.exp_path:
            os.makedirs(args.exp_path, exist_ok=True)
            sd = {"module." + k: v.cpu() for k, v in eng.state_dict().items()}        # DataParallel-style keys (main.py:270)
            for k, v in model.state_dict().items():
                sd.setdefault("module." + k, v)                                       # BatchNorm buffers
            torch.save({"epoch": args.epochs, "arch": args.arch, "state_dict": sd}, os.path.join(args.exp_path, "checkpoint.pth.tar"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
