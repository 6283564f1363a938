"""Packed feature store: packing on the CPU, batch assembly on the GPU; both against the
host mirror of the reference's TSNDataSet (whose segment indices are pinned bit-exact against the reference in
tests/test_index.py)."""
import numpy as np
import pytest
import torch

from ta3n_amd import feature_store
from ta3n_amd.dataset import TSNDataSet


def _make_dataset(tmp_path, D=64, lengths=(1, 2, 3, 4, 5, 6, 9, 17, 40, 101, 250)):
    g = torch.Generator().manual_seed(3)
    lines = []
    for v, n in enumerate(lengths):
        d = tmp_path / f"vid{v}"
        d.mkdir()
        for f in range(1, n + 1):
            torch.save(torch.randn(D, generator=g).abs(), str(d / f"img_{f:05d}.t7"))
        lines.append(f"{d}/ {n} {v % 7}")
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(lines) + "\n")
    return str(lst), D, list(lengths)


def test_pack_layout(tmp_path):
    lst, D, lengths = _make_dataset(tmp_path)
    prefix = str(tmp_path / "packed")
    n, dim = feature_store.pack(lst, prefix)
    assert (n, dim) == (len(lengths), D)
    idx = np.load(prefix + ".idx.npy")
    assert idx[:, 1].tolist() == lengths and idx[:, 2].tolist() == [v % 7 for v in range(len(lengths))]
    assert idx[:, 0].tolist() == np.concatenate(([0], np.cumsum(lengths)[:-1])).tolist()
    blob = np.fromfile(prefix + ".f32", dtype=np.float32).reshape(-1, D)
    assert blob.shape[0] == sum(lengths)
    ds = TSNDataSet("", lst, num_dataload=n, num_segments=5, new_length=1, modality="RGB", test_mode=True)
    x, y = ds[9]                                       # 101 frames: segment ids 11, 31, 51, 71, 91 (dataset.py:103-116)
    rows = idx[9, 0] + np.array([10, 30, 50, 70, 90])
    assert np.array_equal(x.numpy(), blob[rows]) and y == idx[9, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("T", [2, 5, 9])
def test_device_gather_is_bit_exact_with_the_dataset(tmp_path, T):
    lst, D, lengths = _make_dataset(tmp_path)
    prefix = str(tmp_path / "packed")
    n, dim = feature_store.pack(lst, prefix)
    fs = feature_store.FeatureStore(prefix, dim)
    ds = TSNDataSet("", lst, num_dataload=n, num_segments=T, new_length=1, modality="RGB", test_mode=True)
    ids = torch.tensor([10, 0, 3, 3, 9, 1, 2, 4, 5, 6, 7, 8], dtype=torch.int32, device="cuda")
    seg = torch.empty(ids.numel() * T, dtype=torch.int32, device="cuda")
    x, y = fs.gather(ids, T, segment_ids_out=seg)
    torch.cuda.synchronize()
    for k, vid in enumerate(ids.tolist()):
        ref_x, ref_y = ds[vid]
        assert torch.equal(x[k].cpu(), ref_x), (vid, T)
        assert int(y[k]) == ref_y
        assert seg[k * T:(k + 1) * T].tolist() == ds._get_test_indices(ds.video_list[vid])


@pytest.mark.gpu
@pytest.mark.parametrize("T", [3, 5, 9, 12, 25])
def test_device_segment_indices_equal_the_reference_dataset(tmp_path, T):
    """The fp64 index arithmetic of ta3n_gather_segments on the device against tests/golden/index_golden.npz - what the REFERENCE's
    TSNDataSet._get_test_indices returned for 1..400 frames (tests/golden/make_index_golden.py), not a restatement of it."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "index_golden.npz"))[f"segidx_S{T}_L1"]
    lengths = [int(r[0]) for r in gold]
    D = 4
    prefix = str(tmp_path / "packed")
    starts = np.concatenate(([0], np.cumsum(lengths)[:-1]))
    np.save(prefix + ".idx.npy", np.stack([starts, np.asarray(lengths), np.zeros(len(lengths), dtype=np.int64)], axis=1).astype(np.int64))
    blob = np.repeat(np.arange(sum(lengths), dtype=np.float32)[:, None], D, axis=1)      # row r holds the value r
    blob.tofile(prefix + ".f32")
    fs = feature_store.FeatureStore(prefix, D)
    ids = torch.arange(len(lengths), dtype=torch.int32, device="cuda")
    seg = torch.empty(ids.numel() * T, dtype=torch.int32, device="cuda")
    x, _ = fs.gather(ids, T, segment_ids_out=seg)
    torch.cuda.synchronize()
    assert seg.view(-1, T).cpu().tolist() == gold[:, 1:].tolist()
    want_rows = torch.from_numpy(starts[:, None] + gold[:, 1:] - 1).to(torch.float32)      # 1-based frame id -> blob row
    assert torch.equal(x[:, :, 0].cpu(), want_rows)


@pytest.mark.gpu
def test_gather_feeds_the_train_step_input_buffer(tmp_path):
    from ta3n_amd.engine import TrainEngine
    lst, D, lengths = _make_dataset(tmp_path, D=512)
    prefix = str(tmp_path / "packed")
    n, dim = feature_store.pack(lst, prefix)
    fs = feature_store.FeatureStore(prefix, dim)
    eng = TrainEngine(6, 4, 5, 512, 64, 7, dropout_i=0.0, dropout_v=0.0)
    src = torch.tensor([0, 1, 2, 3, 4, 5], dtype=torch.int32, device="cuda")
    tgt = torch.tensor([6, 7, 8, 9], dtype=torch.int32, device="cuda")
    fs.gather(src, 5, out=eng.X[: 6 * 5], labels_out=eng._labels[:6])          # straight into the static buffers
    fs.gather(tgt, 5, out=eng.X[6 * 5:])
    ds = TSNDataSet("", lst, num_dataload=n, num_segments=5, new_length=1, modality="RGB", test_mode=True)
    ref = torch.cat([ds[i][0] for i in range(10)])
    torch.cuda.synchronize()
    assert torch.equal(eng.X.cpu(), ref)
    assert eng._labels[:6].tolist() == [ds[i][1] for i in range(6)]
    for v in eng.param_views().values():
        v.normal_(0, 0.05)
    eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-3)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.P).all() and eng.losses()["loss"] > 0


@pytest.mark.gpu
def test_gather_into_writes_the_input_and_its_bf16_twin(tmp_path):
    """ta3n_gather_segments_into: the batch rows of engine.X bit-exact with the dataset AND, for a twin-reading engine,
    the input twin = RNE_bf16 of those rows, in the same pass."""
    from ta3n_amd.engine import TrainEngine
    lst, D, lengths = _make_dataset(tmp_path, D=512)
    prefix = str(tmp_path / "packed")
    n, dim = feature_store.pack(lst, prefix)
    fs = feature_store.FeatureStore(prefix, dim)
    eng = TrainEngine(6, 4, 5, 512, 64, 7, dropout_i=0.0, dropout_v=0.0, bf16=True, bf16_store=True)
    src = torch.tensor([0, 1, 2, 3, 4, 5], dtype=torch.int32, device="cuda")
    tgt = torch.tensor([6, 7, 8, 9], dtype=torch.int32, device="cuda")
    fs.gather_into(eng, src, 0, labels_out=eng._labels[:6])
    fs.gather_into(eng, tgt, 6)
    ds = TSNDataSet("", lst, num_dataload=n, num_segments=5, new_length=1, modality="RGB", test_mode=True)
    ref = torch.cat([ds[i][0] for i in range(10)])
    torch.cuda.synchronize()
    assert torch.equal(eng.X.cpu(), ref)
    assert eng._labels[:6].tolist() == [ds[i][1] for i in range(6)]
    twin = eng.region("x16").view(torch.int16)[: eng.X.numel()].cpu()
    assert torch.equal(twin, ref.reshape(-1).to(torch.bfloat16).view(torch.int16))
    for v in eng.param_views().values():
        v.normal_(0, 0.05)
    eng.refresh_bf16(params=True)
    eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-3)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.P).all() and eng.losses()["loss"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--use_bn", "AdaBN"], ["--ens_DA", "MCD", "--mu", "0.5"], ["--ens_DA", "MCD", "--mu", "0.5", "--dis_DA", "DAN", "--alpha", "1"],
                                   ["--arithmetic", "f32x3"]],
                         ids=["adabn", "mcd", "mcd_dan", "f32x3_pair_twins"])
def test_train_script_with_the_paper_baseline_options(tmp_path, extra):
    """train_ddp.py on one GPU with --use_bn / --ens_DA / --dis_DA (the native loop of TrainEngine: unfused launch lists + the small
    logit-level losses) and with the split arithmetic on stored hi / lo planes: runs, validates, checkpoints and resumes."""
    import os
    import subprocess
    import sys
    lst, D, lengths = _make_dataset(tmp_path, D=512, lengths=(3, 5, 8, 13, 21, 34, 55, 9, 6, 40, 17, 25))
    prefix = str(tmp_path / "packed")
    feature_store.pack(lst, prefix)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "train_ddp.py"), "no_class_file", "RGB", "a", "b", "c",
            "--frame_aggregation", "trn-m", "--baseline_type", "video", "--arch", "resnet18", "--num_segments", "5",
            "--add_fc", "1", "--fc_dim", "64", "-b", "6", "4", "6", "--lr", "0.01", "--lr_adaptive", "dann",
            "--use_target", "uSv", "--adv_DA", "RevGrad", "--use_attn", "TransAttn", "--add_loss_DA", "attentive_entropy",
            "--place_adv", "Y", "Y", "Y", "--beta", "0.75", "0.75", "0.5", "--gamma", "0.003", "--print_freq", "1",
            "--feature_store", prefix, prefix, prefix, "--exp_path", str(tmp_path / "exp") + "/", "--save_model"] + extra
    r = subprocess.run(base + ["--epochs", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("Train: [") >= 2 and r.stdout.count("Test: [") == 2 and "nan" not in r.stdout.lower(), r.stdout[-2000:]
    if "MCD" in extra:
        assert "loss_s" in r.stdout
    ck = os.path.join(str(tmp_path / "exp"), "RGB", "checkpoint.pth.tar")
    assert os.path.exists(ck)
    sd = torch.load(ck, map_location="cpu", weights_only=False)["state_dict"]
    if "AdaBN" in extra:      # the engine's running statistics, not the initial buffers of the VideoModel it was initialised from
        assert int(sd["module.bn_shared_S.num_batches_tracked"]) > 0 and sd["module.bn_shared_S.running_var"].ne(1).any()
    r2 = subprocess.run(base + ["--epochs", "3", "--resume", ck], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and "=> loaded checkpoint" in r2.stdout and r2.stdout.count("Test: [") == 1, r2.stdout[-2000:] + r2.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("arithmetic", ["f32", "bf16"])
def test_train_script_end_to_end_from_packed_stores(tmp_path, arithmetic):
    """train_ddp.py on one GPU: batches gathered on the device from packed stores, DANN schedules, device-side
    validation - the reference's train/validate loop (main.py:228-274) without its per-frame file reads."""
    import os
    import subprocess
    import sys
    lst, D, lengths = _make_dataset(tmp_path, D=512, lengths=(3, 5, 8, 13, 21, 34, 55, 9, 6, 40, 17, 25))
    prefix = str(tmp_path / "packed")
    feature_store.pack(lst, prefix)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "train_ddp.py"), "no_class_file", "RGB", "a", "b", "c",
           "--frame_aggregation", "trn-m", "--baseline_type", "video", "--arch", "resnet18", "--num_segments", "5",
           "--add_fc", "1", "--fc_dim", "64", "-b", "6", "4", "6", "--epochs", "3", "--lr", "0.01", "--lr_adaptive", "dann",
           "--use_target", "uSv", "--adv_DA", "RevGrad", "--use_attn", "TransAttn", "--add_loss_DA", "attentive_entropy",
           "--place_adv", "Y", "Y", "Y", "--beta", "0.75", "0.75", "0.5", "--gamma", "0.003", "--print_freq", "1",
           "--feature_store", prefix, prefix, prefix, "--arithmetic", arithmetic]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("Train: [") >= 3 and r.stdout.count("Test: [") == 3, r.stdout[-2000:]
    assert "nan" not in r.stdout.lower()


def test_pack_bf16_layout(tmp_path):
    """dtype='bf16': the same rows rounded to bf16 (nearest even), raw 2-byte elements."""
    lst, D, lengths = _make_dataset(tmp_path)
    n, dim = feature_store.pack(lst, str(tmp_path / "p32"))
    n2, dim2 = feature_store.pack(lst, str(tmp_path / "p16"), dtype="bf16")
    assert (n, dim) == (n2, dim2)
    b32 = torch.from_numpy(np.fromfile(str(tmp_path / "p32.f32"), dtype=np.float32))
    b16 = torch.from_numpy(np.fromfile(str(tmp_path / "p16.bf16"), dtype=np.int16))
    assert torch.equal(b16, b32.to(torch.bfloat16).view(torch.int16))
    assert np.array_equal(np.load(str(tmp_path / "p16.idx.npy")), np.load(str(tmp_path / "p32.idx.npy")))


@pytest.mark.gpu
def test_bf16_store_feeds_the_input_twin_directly(tmp_path):
    """A bf16 store assembles the batch into the input's bf16 twin without touching fp32 (2 B in + 2 B out per element): the
    twin equals the one the fp32 store + conversion path builds, bit for bit, and a train step from either is the same step."""
    from ta3n_amd.engine import TrainEngine
    lst, D, lengths = _make_dataset(tmp_path, D=64)
    feature_store.pack(lst, str(tmp_path / "p32"))
    feature_store.pack(lst, str(tmp_path / "p16"), dtype="bf16")
    s32 = feature_store.FeatureStore(str(tmp_path / "p32"), D)
    s16 = feature_store.FeatureStore(str(tmp_path / "p16"), D)
    assert s16.bf16 and not s32.bf16
    T, Bs, Bt = 5, 6, 5
    ids = torch.arange(len(lengths), dtype=torch.int32)
    res = []
    for store in (s32, s16):
        eng = TrainEngine(Bs, Bt, T, D, 32, 7, dropout_i=0.0, dropout_v=0.0, bf16=True, bf16_store=True)
        for v in eng.param_views().values():
            v.normal_(0, 0.05, generator=torch.Generator(device="cuda").manual_seed(1))
        eng.refresh_bf16(params=True)
        store.gather_into(eng, ids[:Bs].cuda(), 0, labels_out=eng._labels[:Bs])
        store.gather_into(eng, ids[Bs:Bs + Bt].cuda(), Bs)
        eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-2)
        torch.cuda.synchronize()
        res.append((eng.region("x16").clone(), eng.P.clone(), eng._labels[:Bs].clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2])
    assert torch.equal(res[0][1], res[1][1]) and res[0][1].abs().max().item() > 0
    # the host-side gather of a bf16 store widens the same rows
    f32, lab32 = s32.gather(ids[:4].cuda(), T)
    f16, lab16 = s16.gather(ids[:4].cuda(), T)
    assert torch.equal(f16, f32.to(torch.bfloat16).to(torch.float32)) and torch.equal(lab32, lab16)
    # an engine that does NOT read twins gets widened fp32 rows from the bf16 store
    eng = TrainEngine(Bs, Bt, T, D, 32, 7, dropout_i=0.0, dropout_v=0.0)
    s16.gather_into(eng, ids[:Bs].cuda(), 0)
    torch.cuda.synchronize()
    assert torch.equal(eng.X[: Bs * T].view(Bs, T, D), s16.gather(ids[:Bs].cuda(), T)[0])


@pytest.mark.gpu
def test_train_script_chunked_steps_log_the_same_numbers(tmp_path):
    """train_ddp.py sends the steps between two log lines to the GPU in one call (TrainEngine.train_steps with device-side batch
    feeds); with TA3N_TRAIN_CHUNKS=0 it makes one call per step.  Same arithmetic: the logged losses and the validation results
    must be identical, line by line."""
    import os
    import subprocess
    import sys
    lst, D, lengths = _make_dataset(tmp_path, D=512, lengths=(3, 5, 8, 13, 21, 34, 55, 9, 6, 40, 17, 25, 11, 4, 30, 7, 19, 23))
    prefix = str(tmp_path / "packed")
    feature_store.pack(lst, prefix)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "train_ddp.py"), "no_class_file", "RGB", "a", "b", "c",
           "--frame_aggregation", "trn-m", "--baseline_type", "video", "--arch", "resnet18", "--num_segments", "5",
           "--add_fc", "1", "--fc_dim", "64", "-b", "6", "3", "6", "--epochs", "2", "--lr", "0.01", "--lr_adaptive", "dann",
           "--use_target", "uSv", "--adv_DA", "RevGrad", "--use_attn", "TransAttn", "--add_loss_DA", "attentive_entropy",
           "--place_adv", "Y", "Y", "Y", "--beta", "0.75", "0.75", "0.5", "--gamma", "0.003", "--print_freq", "2",
           "--dropout_i", "0", "--dropout_v", "0", "--feature_store", prefix, prefix, prefix, "--arithmetic", "f32"]
    outs = []
    for chunks in ("1", "0"):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, TA3N_TRAIN_CHUNKS=chunks))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith(("Train: [", "Test: ["))])
    assert len(outs[0]) >= 4 and outs[0] == outs[1], "\n".join(outs[0] + ["---"] + outs[1])
