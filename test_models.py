#!/usr/bin/env python
"""Video-level testing of a trained checkpoint: the reference's test_models.py (same command line, same output lines) on the
HIP-backed ta3n_amd.models.VideoModel.

    python test_models.py data/classInd_hmdb_ucf.txt RGB <test_list> <exp>/RGB/model_best.pth.tar \\
        --arch resnet101 --test_segments 5 --baseline_type video --frame_aggregation trn-m --use_attn TransAttn --bS 128 \\
        [--save_confusion <prefix>] [--save_scores <prefix>] [--save_attention <prefix>]

Reference lines cited as :N are cmhungsteve/TA3N test_models.py."""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from ta3n_amd.dataset import TSNDataSet  # noqa: E402
from ta3n_amd.models import VideoModel  # noqa: E402
from ta3n_amd.utils.utils import plot_confusion_matrix  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="Standard video-level testing")                  # :24-71
    p.add_argument('class_file', type=str, default="classInd.txt")
    p.add_argument('modality', type=str, choices=['RGB', 'Flow', 'RGBDiff', 'RGBDiff2', 'RGBDiffplus'])
    p.add_argument('test_list', type=str)
    p.add_argument('weights', type=str)
    p.add_argument('--arch', type=str, default="resnet101")
    p.add_argument('--test_segments', type=int, default=5)
    p.add_argument('--add_fc', default=1, type=int)
    p.add_argument('--fc_dim', type=int, default=512)
    p.add_argument('--baseline_type', type=str, default='frame', choices=['frame', 'video', 'tsn'])
    p.add_argument('--frame_aggregation', type=str, default='avgpool', choices=['avgpool', 'rnn', 'temconv', 'trn-m', 'none'])
    p.add_argument('--dropout_i', type=float, default=0)
    p.add_argument('--dropout_v', type=float, default=0)
    p.add_argument('--n_rnn', default=1, type=int)
    p.add_argument('--rnn_cell', type=str, default='LSTM', choices=['LSTM', 'GRU'])
    p.add_argument('--n_directions', type=int, default=1, choices=[1, 2])
    p.add_argument('--n_ts', type=int, default=5)
    p.add_argument('--share_params', type=str, default='Y', choices=['Y', 'N'])
    p.add_argument('--use_bn', type=str, default='none', choices=['none', 'AdaBN', 'AutoDIAL'])
    p.add_argument('--use_attn_frame', type=str, default='none', choices=['none', 'TransAttn', 'general', 'DotProduct'])
    p.add_argument('--use_attn', type=str, default='none', choices=['none', 'TransAttn', 'general', 'DotProduct'])
    p.add_argument('--n_attn', type=int, default=1)
    p.add_argument('--ens_DA', type=str, default='none', choices=['none', 'MCD'],
                   help="not in the reference's tester: its strict load_state_dict cannot read an MCD checkpoint (the second classifier)")
    p.add_argument('--top', default=[1, 3, 5], nargs='+', type=int)
    p.add_argument('--verbose', default=False, action="store_true")
    p.add_argument('--save_confusion', type=str, default=None)
    p.add_argument('--save_scores', type=str, default=None)
    p.add_argument('--save_attention', type=str, default=None)
    p.add_argument('--max_num', type=int, default=-1)
    p.add_argument('-j', '--workers', default=4, type=int)
    p.add_argument('--bS', default=2, type=int)
    p.add_argument('--gpus', nargs='+', type=int, default=None)
    p.add_argument('--flow_prefix', type=str, default='')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    class_names = [line.strip().split(' ', 1)[1] for line in open(args.class_file)]        # :73-74
    num_class = len(class_names)
    top = sorted(args.top)
    kmax = min(max(top), num_class)
    print('preparing the model......')                                                       # :77-85
    seg = args.test_segments if args.baseline_type == 'video' else 1
    net = VideoModel(num_class, args.baseline_type, args.frame_aggregation, args.modality, train_segments=seg, val_segments=seg,
                     base_model=args.arch, add_fc=args.add_fc, fc_dim=args.fc_dim, share_params=args.share_params,
                     dropout_i=args.dropout_i, dropout_v=args.dropout_v, use_bn=args.use_bn, ens_DA=args.ens_DA, partial_bn=False,
                     n_rnn=args.n_rnn, rnn_cell=args.rnn_cell, n_directions=args.n_directions, n_ts=args.n_ts, use_attn=args.use_attn,
                     n_attn=args.n_attn, use_attn_frame=args.use_attn_frame, verbose=args.verbose)
    checkpoint = torch.load(args.weights, map_location="cpu", weights_only=False)            # :87-92
    print("model epoch {} prec@1: {}".format(checkpoint['epoch'], checkpoint['prec1']))
    net.load_state_dict({'.'.join(k.split('.')[1:]): v for k, v in list(checkpoint['state_dict'].items())})
    print('loading data......')                                                              # :95-106
    data_length = 1 if args.modality == "RGB" else 5
    num_test = sum(1 for _ in open(args.test_list))
    tmpl = "img_{:05d}.t7" if args.modality in ['RGB', 'RGBDiff', 'RGBDiff2', 'RGBDiffplus'] else args.flow_prefix + "{}_{:05d}.t7"
    data_set = TSNDataSet("", args.test_list, num_dataload=num_test, num_segments=args.test_segments, new_length=data_length,
                          modality=args.modality, image_tmpl=tmpl, test_mode=True)
    loader = torch.utils.data.DataLoader(data_set, batch_size=args.bS, shuffle=False, num_workers=args.workers, pin_memory=True)
    net = torch.nn.DataParallel(net.cuda(), device_ids=[0])                                  # :111-112
    net.eval()
    max_num = args.max_num if args.max_num > 0 else len(loader.dataset)
    hits = torch.zeros(len(top), dtype=torch.long)
    total = 0
    confusion = torch.zeros(kmax, num_class, num_class, dtype=torch.long)                    # [k][true][k-th prediction]
    scores, attn_values = [], []
    print('start testing......')
    t0 = time.time()
    for i, (data, label) in enumerate(loader):                                               # :147-186
        if i >= max_num:
            break
        n = data.size(0)
        if n < args.bS:                                                                      # pad the last batch (:149-153)
            data = torch.cat((data, torch.zeros(args.bS - n, data.size(1), data.size(2))))
        with torch.no_grad():                                                                # :126-127
            _, _, _, _, _, attn, out, _, _, _ = net(data.cuda(), data.cuda(), [0, 0, 0], 0, is_train=False, reverse=False)
        prob = nn.Softmax(dim=1)(out[:n]).cpu()
        pred = prob.topk(kmax)[1]                                                            # [n, kmax]
        label = label[:n]
        for j, t in enumerate(top):
            hits[j] += (pred[:, :min(t, kmax)] == label[:, None]).sum()
        for k in range(kmax):
            confusion[k].index_put_((label, pred[:, k]), torch.ones(n, dtype=torch.long), accumulate=True)
        total += n
        scores.append(prob.numpy())
        attn_values.append(attn[:n].detach().cpu())
        if args.verbose or i == len(loader) - 1:
            line = ''.join('Pred@%d %f, ' % (t, float(hits[j]) / total) for j, t in enumerate(top))
            print(line + 'average %f sec/video' % ((time.time() - t0) / total))
    if args.save_attention:                                                                  # :188-189
        np.savetxt(args.save_attention + '.txt', torch.cat(attn_values).reshape(total, -1).numpy(), fmt="%s")
    if args.save_scores:
        np.save(args.save_scores + '.npy', np.concatenate(scores))
    cf = confusion.numpy()                                                                   # :191-199
    cls_cnt = cf[0].sum(axis=1)
    cls_hit = np.array([np.diag(cf[k]) for k in range(kmax)])
    cls_acc_topK = [cls_hit[:min(t, kmax)].sum(axis=0) / np.maximum(cls_cnt, 1) for t in top]
    if args.save_confusion:
        plot_confusion_matrix(args.save_confusion + '.png', cf[0], classes=class_names, normalize=True, title='Normalized confusion matrix')
        with open(args.save_confusion + '-top' + str(top) + '.txt', 'w') as f:              # :214-224
            for c in range(num_class):
                f.write(' '.join(str(a[c]) for a in cls_acc_topK) + ' \n')
    if args.verbose:
        for c in range(num_class):
            print(' '.join(str(a[c]) for a in cls_acc_topK))
    final = ''.join('Pred@{:d} {:.02f}% '.format(t, cls_hit[:min(t, kmax)].sum() / max(cls_cnt.sum(), 1) * 100) for t in top)   # :208-211
    print(final)
    return {t: float(cls_hit[:min(t, kmax)].sum() / max(cls_cnt.sum(), 1) * 100) for t in top}


if __name__ == "__main__":
    main()
