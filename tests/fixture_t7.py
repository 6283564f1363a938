"""Generates a tiny dataset in the reference's on-disk format (README.md:62-95, dataset.py:53-60): one directory per
video with img_00001.t7 ... (a 1-D float tensor per frame, torch.save), list files "<dir>/ <num_frames> <label>", a class
file.  Features are half-normal with a class-dependent mean shift so that a few epochs visibly learn."""
import os

import torch


def make_dataset(root, n_class=5, dim=512, videos=(24, 20, 12), frames=(7, 12), seed=0):
    """Returns (class_file, source_list, target_list, val_list)."""
    g = torch.Generator().manual_seed(seed)
    os.makedirs(root, exist_ok=True)
    class_file = os.path.join(root, "classInd.txt")
    with open(class_file, "w") as f:
        for c in range(n_class):
            f.write(f"{c} class{c}\n")
    proto = torch.randn(n_class, dim, generator=g)
    lists = []
    for split, n in zip(("source", "target", "val"), videos):
        path = os.path.join(root, f"list_{split}.txt")
        with open(path, "w") as lf:
            for v in range(n):
                label = v % n_class
                nf = int(torch.randint(frames[0], frames[1] + 1, (1,), generator=g))
                d = os.path.join(root, split, f"v{v:03d}")
                os.makedirs(d, exist_ok=True)
                shift = 0.0 if split == "source" else 0.3            # a domain gap
                for k in range(1, nf + 1):
                    feat = (torch.randn(dim, generator=g) + 1.5 * proto[label] + shift).abs()
                    torch.save(feat, os.path.join(d, "img_{:05d}.t7".format(k)))
                lf.write(f"{d}/ {nf} {label}\n")
        lists.append(path)
    return (class_file, *lists)
