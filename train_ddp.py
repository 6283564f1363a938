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
Schedules follow main.py: beta (main.py:350-352), learning rate (DANN main.py:620-621, step decay :236-237), the
list-repeat rule of --copy_list (main.py:145-153), checkpoints in the reference's format (ta3n_amd/checkpoint.py).
Every option value the engine does not implement is REJECTED at start-up (validate_options) instead of being ignored."""
import math
import os
import sys
import time

# dmabuf IPC for cross-process device memory (RCCL ranks, the opt-in peer transport): must be in the environment before the HIP runtime
# initialises, i.e. before `import torch` (only tests/conftest.py used to set it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ta3n_amd import checkpoint as ckpt  # noqa: E402
from ta3n_amd import parallel  # noqa: E402
from ta3n_amd.engine import TrainEngine, beta_dann, flags_from_options, lr_dann  # noqa: E402
from ta3n_amd.models import ARCH_FEATURE_DIM  # noqa: E402
from ta3n_amd.opts import parser  # noqa: E402


def validate_options(args, module_path: bool = False) -> None:
    """The engine implements the TA3N hot path and BASELINE configs[0]; every behaviour-changing value
    outside it stops the run here - a silently ignored flag would train a different model than the command line says.
    module_path: the caller assembles the loss from VideoModel.forward's tensors (main.py) - there the discrepancy losses
    (--dis_DA DAN / JAN) and --ens_DA MCD are built as well."""
    bad = []

    def need(cond, msg):
        if not cond:
            bad.append(msg)
    need(args.baseline_type == "video", f"--baseline_type {args.baseline_type} (built: video)")
    need(args.frame_aggregation in ("trn-m", "avgpool"), f"--frame_aggregation {args.frame_aggregation} (built: trn-m, avgpool)")
    if args.frame_aggregation == "avgpool":      # TemPooling: source-only (BASELINE configs[0]) or with the RevGrad branches; no attention
        need(args.use_attn == "none" and (args.add_loss_DA == "none" or args.use_target == "none"),
             "avgpool is built without attention / attentive entropy (--use_attn none --add_loss_DA none, as the reference's script runs it)")
    need(args.optimizer == "SGD", f"--optimizer {args.optimizer} (built: SGD with Nesterov momentum, main.py:83)")
    if module_path:
        need(args.dis_DA in ("none", "DAN", "JAN"), f"--dis_DA {args.dis_DA} (built: DAN, JAN)")
        need(args.ens_DA in ("none", "MCD"), f"--ens_DA {args.ens_DA}")
        if args.dis_DA == "DAN":
            need(len(args.place_dis) == args.add_fc + 2 and args.place_dis[2] == "N",
                 "--place_dis takes add_fc + 2 values [logits, video feature, frame features]; the reference itself fails on the "
                 "frame features (loss.py:49 on a 3-D tensor)")
    else:
        need(args.dis_DA in ("none", "DAN", "JAN"), f"--dis_DA {args.dis_DA} (built: DAN, JAN)")
        need(args.ens_DA in ("none", "MCD"), f"--ens_DA {args.ens_DA}")
        if args.ens_DA == "MCD":
            need(args.use_bn == "none", "--ens_DA MCD with --use_bn (use main.py, the module path)")
    if module_path:
        need(args.use_bn in ("none", "AdaBN", "AutoDIAL"), f"--use_bn {args.use_bn}")
    else:
        need(args.use_bn in ("none", "AdaBN", "AutoDIAL"), f"--use_bn {args.use_bn}")
    need(args.add_loss_DA in ("none", "attentive_entropy"), f"--add_loss_DA {args.add_loss_DA} (built: attentive_entropy)")
    need(args.use_target in ("none", "uSv"), f"--use_target {args.use_target} (target labels in the classification loss are not built)")
    need(args.weighted_class_loss == "N", "--weighted_class_loss Y")
    need(args.weighted_class_loss_DA == "N", "--weighted_class_loss_DA Y")
    need(args.pred_normalize == "N", "--pred_normalize Y")
    need(not args.pretrain_source, "--pretrain_source")
    need(args.lr_adaptive in ("dann", "none"), f"--lr_adaptive {args.lr_adaptive} (built: dann, none with --lr_steps/--lr_decay)")
    need(args.use_attn in ("TransAttn", "none"), f"--use_attn {args.use_attn}")
    need(args.use_attn_frame == "none", f"--use_attn_frame {args.use_attn_frame}")
    need(args.share_params == "Y", "--share_params N")
    need(args.add_fc == 1, f"--add_fc {args.add_fc}")
    need(args.modality == "RGB", f"modality {args.modality} (pre-extracted RGB features)")
    need(args.mu == 0 or args.ens_DA == "MCD", f"--mu {args.mu} (only used by --ens_DA MCD)")
    need(len(args.beta) == 3 and len(args.place_adv) == 3, "--beta and --place_adv take three values [relation, video, frame]")
    need(len(args.batch_size) >= 2, "-b needs at least the source and target batch sizes")
    need(args.arch in ARCH_FEATURE_DIM, f"--arch {args.arch}")
    if args.add_loss_DA == "attentive_entropy" and args.use_attn != "none" and args.use_target != "none":
        need(args.place_adv[0] == "Y" and args.place_adv[1] == "Y",
             "attentive_entropy indexes pred_domain_all[1] (main.py:559-562): needs --place_adv Y Y *")
    if bad:
        raise SystemExit("train_ddp.py: unsupported option value(s):\n  " + "\n  ".join(bad))


def train_list_sizes(num_source: int, num_target: int, batch_size, copy_list):
    """main.py:145-153: the shorter list is repeated (--copy_list Y) so that both loaders run the same number of
    iterations.  Returns (num_source_train, num_target_train)."""
    num_iter_source = num_source / batch_size[0]
    num_iter_target = num_target / batch_size[1]
    num_max_iter = max(num_iter_source, num_iter_target)
    ns = round(num_max_iter * batch_size[0]) if copy_list[0] == "Y" else num_source
    nt = round(num_max_iter * batch_size[1]) if copy_list[1] == "Y" else num_target
    return ns, nt


def n_batches(n: int, batch: int) -> int:
    return max(-(-n // batch), 1)        # DataLoader(drop_last=False)


def synthetic_loader(n_videos, batch, T, D, C, seed):
    """Batches of half-normal features [b,T,D] + labels, like TSNDataSet items stacked by a DataLoader."""
    g = torch.Generator().manual_seed(seed)
    left = n_videos
    while left > 0:
        b = min(batch, left)
        yield torch.randn(b, T, D, generator=g).abs_(), torch.randint(0, C, (b,), generator=g)
        left -= b


def list_loader(list_file, n_load, batch, T, seed):
    from ta3n_amd.dataset import TSNDataSet
    ds = TSNDataSet("", list_file, num_dataload=n_load, num_segments=T, new_length=1, modality="RGB", random_shift=False,
                    test_mode=True)
    g = torch.Generator().manual_seed(seed)             # same permutation on every rank: shards are disjoint slices
    sampler = torch.utils.data.RandomSampler(ds, generator=g)
    return torch.utils.data.DataLoader(ds, batch_size=batch, sampler=sampler, num_workers=0, drop_last=False)


def main():
    parser.add_argument("--synthetic", type=int, nargs=2, default=None, metavar=("N_SRC", "N_TGT"),
                        help="train on synthetic features instead of the list files")
    parser.add_argument("--arithmetic", choices=("f32", "bf16", "f32x3"), default="f32",
                        help="f32: the reference's arithmetic (fp32 MFMA); bf16: contraction operands rounded to bf16 and read from bf16 "
                             "twins, fp32 accumulation / parameters / optimiser (BASELINE configs[1]); f32x3: fp32-grade contractions as "
                             "three bf16 MFMAs on operands split hi + lo (meets the fp32 parity bounds, ~25 %% faster than f32)")
    parser.add_argument("--graph", action="store_true", help="replay a captured hipGraph (default: eager launches, faster here)")
    parser.add_argument("--feature_store", type=str, nargs="+", default=None, metavar="PREFIX",
                        help="packed feature stores (ta3n_amd.feature_store.pack): SRC TGT [VAL]; batches are assembled on the GPU")
    args = parser.parse_args()
    validate_options(args)
    rank, local_rank, world = parallel.init_distributed()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    num_class = len([x for x in open(args.class_file)]) if os.path.exists(args.class_file) else 12
    T, D = args.num_segments, ARCH_FEATURE_DIM[args.arch]
    Bs_g, Bt_g = args.batch_size[0], args.batch_size[1]
    Bs, Bt = parallel.padded_shard_size(Bs_g, world), parallel.padded_shard_size(Bt_g, world)
    flags = flags_from_options(args.place_adv, args.add_loss_DA, args.use_attn, args.adv_DA, args.use_target)
    eng = TrainEngine(Bs, Bt, T, D, args.fc_dim, num_class, flags=flags, dropout_i=args.dropout_i,
                      dropout_v=args.dropout_v, momentum=args.momentum, weight_decay=args.weight_decay,
                      clip=args.clip_gradient, device=dev, bf16=(args.arithmetic == "bf16"), bf16_store=(args.arithmetic == "bf16"),
                      f32_split=(args.arithmetic == "f32x3"), aggregation=args.frame_aggregation,
                      dis_DA=args.dis_DA, place_dis=args.place_dis, alpha=max(args.alpha, 0.0), use_bn=args.use_bn,
                      ens_DA=args.ens_DA, mu=args.mu)
    from ta3n_amd.models import VideoModel
    torch.manual_seed(1)
    model = VideoModel(num_class, args.baseline_type, args.frame_aggregation, args.modality, train_segments=T,
                       val_segments=T, base_model=args.arch, add_fc=args.add_fc, fc_dim=args.fc_dim,
                       dropout_i=args.dropout_i, dropout_v=args.dropout_v, partial_bn=not args.no_partialbn,
                       use_bn=args.use_bn, ens_DA=args.ens_DA, use_attn=args.use_attn, verbose=False)
    eng.load_state(model.state_dict())                  # reference initialisation under torch.manual_seed(1)
    start_epoch, best_prec1, lr_resumed = 1, 0.0, None
    if args.resume:                                      # main.py:94-106
        if not os.path.isfile(args.resume):
            raise SystemExit(f"=> no checkpoint found at '{args.resume}'")
        st = ckpt.load_into_engine(eng, model, torch.load(args.resume, map_location="cpu", weights_only=False), args.resume_hp)
        start_epoch, best_prec1, lr_resumed = st["start_epoch"], st["best_prec1"], st["lr"]
        if rank == 0:
            print(f"=> loaded checkpoint '{args.resume}' (epoch {start_epoch - 1})", flush=True)
    parallel.broadcast_(eng.P)
    parallel.broadcast_(eng.M)
    eng.refresh_bf16(params=True)
    stores = None
    if args.feature_store:
        from ta3n_amd.feature_store import FeatureStore
        stores = [FeatureStore(pfx, D, dev) for pfx in args.feature_store]
        n_src, n_tgt = len(stores[0]), len(stores[1])
    elif args.synthetic:
        n_src, n_tgt = args.synthetic
    else:
        n_src, n_tgt = sum(1 for _ in open(args.train_source_list)), sum(1 for _ in open(args.train_target_list))
    n_src_train, n_tgt_train = train_list_sizes(n_src, n_tgt, args.batch_size, args.copy_list)
    len_source_loader = n_batches(n_src_train, Bs_g)                      # main.py:334-335 use len(source_loader)
    steps_per_epoch = min(len_source_loader, n_batches(n_tgt_train, Bt_g))   # zip() stops at the shorter loader (main.py:348)
    captured = False

    def store_loader(n_videos, n_load, batch, seed):
        """Video ids of one epoch: a seeded permutation of the list repeated to n_load entries (dataset.py:69-74 +
        RandomSampler, main.py:176-190), identical on every rank."""
        ids = torch.arange(n_load) % n_videos
        perm = ids[torch.randperm(n_load, generator=torch.Generator().manual_seed(seed))]
        for b in range(n_batches(n_load, batch)):
            yield perm[b * batch:(b + 1) * batch].to(torch.int32)

    def validate(epoch):
        """main.validate (main.py:669-761): eval-mode forward + device-side CE / top-k / confusion matrix (same on every rank)."""
        val = stores[2]
        for b0 in range(0, len(val), Bs):
            ids = torch.arange(b0, min(b0 + Bs, len(val)), dtype=torch.int32, device=dev)
            x, y = val.gather(ids, T)
            eng.evaluate_batch(x, y, reset=(b0 == 0))
        r = eng.eval_results()
        if rank == 0:
            print(f"Test: [{epoch}] Prec@1 {r['prec1']:.3f} Prec@5 {r['prec5']:.3f} Loss {r['loss']:.5f} ({r['n']} videos)", flush=True)
        return r["prec1"]

    lr = args.lr if lr_resumed is None else lr_resumed
    t_start = time.time()
    for epoch in range(start_epoch, args.epochs + 1):
        if args.lr_adaptive == "none" and epoch in args.lr_steps:                     # main.py:236-237, 790-793
            lr /= args.lr_decay
        # main.py:233: the discrepancy-loss weight follows the epoch when --alpha is negative
        eng.alpha = 2 / (1 + math.exp(-1 * epoch / args.epochs)) - 1 if args.alpha < 0 else args.alpha
        if stores:
            src = ((ids, None) for ids in store_loader(n_src, n_src_train, Bs_g, 1000 + epoch))
            tgt = ((ids, None) for ids in store_loader(max(n_tgt, 1), n_tgt_train, Bt_g, 2000 + epoch))
        elif args.synthetic:
            src = synthetic_loader(n_src_train, Bs_g, T, D, num_class, 1000 + epoch)
            tgt = synthetic_loader(n_tgt_train, Bt_g, T, D, num_class, 2000 + epoch)
        else:
            src = list_loader(args.train_source_list, n_src_train, Bs_g, T, 1000 + epoch)
            tgt = list_loader(args.train_target_list, n_tgt_train, Bt_g, T, 2000 + epoch)
        # With packed stores and full, evenly sharded batches the steps between two log lines go to the GPU in ONE call
        # (TrainEngine.train_steps: the library gathers each step's batch on the device, runs the step, exchanges the gradients and
        # opens the next step with the update) - same arithmetic as one call per step, bit for bit.
        # (when the engine has no single library call for its configuration - torch.distributed fallback, TA3N_DDP_BUCKETS=2,
        # TA3N_SIDE_UPDATE=0 - train_steps itself runs the chunk step by step, gathering each batch on the device: ADVICE r03)
        chunked = (bool(stores) and eng.fused and not args.graph and Bs_g % world == 0 and Bt_g % world == 0 and
                   os.environ.get("TA3N_TRAIN_CHUNKS", "1") == "1")
        chunk = []

        def log_line(i, lr_used, beta):
            # one host sync every print_freq steps (the reference syncs 5-6x per step).  A rank's loss scalars are its
            # shard's sums divided by the GLOBAL counts: the job's losses are their sum over ranks.
            eng.check_exchange()                 # (peer all-reduce only: a rank that gave up waiting is reported here, not silently ignored)
            # (the MCD terms the engine keeps outside the loss kernel are partial sums too; the discrepancy loss is the global value on every rank)
            zero = eng.ws.new_zeros(())
            es = eng.loss_e_shift if eng.loss_e_shift is not None else (zero, zero)
            scal = torch.cat((eng.region("losses")[:6], torch.stack([eng.loss_s if eng.loss_s is not None else zero,
                                                                     eng.loss_c2 if eng.loss_c2 is not None else zero,
                                                                     es[0], es[1]]).to(torch.float32)))
            if world > 1:
                torch.distributed.all_reduce(scal)
            if rank == 0:
                v = scal.tolist()
                extra = ""      # the terms the loss kernel does not know (one rank only): main.py's loss_d / loss_s - printed AND, like in
                # the reference's log (main.py:564-571: `loss` is everything that was backpropagated), part of the logged total
                if eng.loss_d is not None:
                    extra += f" loss_d {eng.loss_d.item():.4f}"
                    v[0] += eng.alpha * eng.loss_d.item()
                if eng.loss_s is not None:
                    extra += f" loss_s {v[6]:.4f} loss_c2 {v[7]:.4f}"
                    v[0] += v[6] + v[7]
                    v[1] += v[7]                        # main.py:447-450: loss_c is the sum of the two classifiers' cross-entropies
                if eng.loss_e_shift is not None:        # MCD: the target rows' entropy term is the SECOND pass's (main.py:549, 559-562)
                    v[0] += v[8]
                    v[5] += v[9]
                print(f"Train: [{epoch}][{i}/{steps_per_epoch}] lr {lr_used:.5f} loss {v[0]:.4f} loss_c {v[1]:.4f} "
                      f"loss_a {v[2] + v[3] + v[4]:.4f} loss_e {v[5]:.4f}{extra} beta {beta[0]:.3f},{beta[1]:.3f},{beta[2]:.3f}", flush=True)

        def flush_chunk():
            if not chunk:
                return
            ids_s = torch.stack([c[3] for c in chunk]).to(dev)
            ids_t = torch.stack([c[4] for c in chunk]).to(dev)
            eng.train_steps([(c[2], args.gamma, c[1]) for c in chunk], feeds=((stores[0], ids_s), (stores[1], ids_t)))
            i_last, lr_last, beta_last = chunk[-1][0], chunk[-1][1], chunk[-1][2]
            chunk.clear()
            if i_last % max(args.print_freq, 1) == 0:
                log_line(i_last, lr_last, beta_last)

        for i, ((xs, ys), (xt, _)) in enumerate(zip(src, tgt)):
            p = float(i + epoch * len_source_loader) / (args.epochs * len_source_loader)   # main.py:334-335, 350
            if i == 0:      # main.py:352 rebinds `beta` inside the loop: a negative entry is replaced by the DANN value of the
                beta = [beta_dann(p) if b < 0 else b for b in args.beta]              # epoch's first step and stays there for the epoch
            lo, hi = parallel.shard_range(xs.size(0), world, rank)                    # this rank's videos
            lo_t, hi_t = parallel.shard_range(xt.size(0), world, rank)
            if chunked and xs.size(0) == Bs_g and xt.size(0) == Bt_g:
                chunk.append((i, lr, list(beta), xs[lo:hi].to(torch.int32), xt[lo_t:hi_t].to(torch.int32)))
                if args.lr_adaptive == "dann":
                    lr = lr_dann(args.lr, p)                                          # main.py:620-621 (takes effect at the next step)
                if i % max(args.print_freq, 1) == 0:
                    flush_chunk()
                continue
            flush_chunk()                                                             # a ragged (last) batch: one step the plain way
            eng.flush()
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
            if not captured and args.graph:
                eng.set_hyper(beta, args.gamma, lr)
                eng.capture()
                captured = True
            eng.train_step(beta, args.gamma, lr, valid_source=hi - lo, valid_target=hi_t - lo_t,
                           global_source=xs.size(0), global_target=xt.size(0))
            lr_used = lr
            if args.lr_adaptive == "dann":
                lr = lr_dann(args.lr, p)                                              # main.py:620-621 (takes effect at the next step)
            if i % max(args.print_freq, 1) == 0:
                log_line(i, lr_used, beta)
        flush_chunk()
        eng.flush()                                                                   # the epoch's last update, before validation / checkpoint
        eng.check_exchange()
        eng.sync_buffers()                                                            # use_bn: every rank evaluates and saves replica 0's running statistics
        if epoch % max(args.eval_freq, 1) == 0 or epoch == args.epochs:                   # main.py:252-274
            prec1 = validate(epoch) if (stores and len(stores) > 2) else 0.0
            is_best = prec1 > best_prec1
            best_prec1 = max(prec1, best_prec1)
            if rank == 0 and args.save_model and args.exp_path:
                ckpt.save_checkpoint(ckpt.engine_checkpoint(eng, model, epoch, args.arch, lr, best_prec1, prec1), is_best,
                                     os.path.join(args.exp_path, args.modality))
    torch.cuda.synchronize(dev)
    if rank == 0:
        print(f"total training time: {time.time() - t_start:.1f}s")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
