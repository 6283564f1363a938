import os
import sys

# Cross-process device-memory sharing (RCCL ranks, the opt-in peer transport) needs dmabuf IPC on these hosts; the variable has to be
# in the environment BEFORE the HIP runtime initialises - i.e. before the first `import torch` of this process and of every process it
# spawns (spawned workers inherit this environment).  Round 4's driver run lost 41 tests to a worker that set it after `import torch`.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the round-end driver): the SURVEY.md 8 rows")
    config.addinivalue_line("markers", "gpu_ab: needs a GPU AND the experiments build (TA3N_LIBDIR=ta3n_amd/lib_ab built with "
                                       "-DTA3N_EXPERIMENTS=1): measured-and-rejected variants whose device code is ABSENT from the default library; "
                                       "never part of `-m gpu` or `-m 'not gpu'`, select with `-m gpu_ab`")


# `-m gpu` runs in this order (files not listed keep their alphabetical place behind these): the parity tests of SURVEY.md 8(a) first,
# then the one-call training loop the benchmark times, the other BASELINE configurations, the RCCL exchange of 8(e), the reference's
# own main.py on the GPU (8b), then module / loader / validation / option coverage (8f).  With `-x` a failure late in the list can no
# longer hide the rows in front of it.
ORDER = [
    "test_gpu_parity.py", "test_gpu_gradients.py", "test_gpu_masked_gradients.py", "test_gpu_bf16.py", "test_gpu_train_steps.py",
    "test_gpu_two_stream.py", "test_gpu_rccl.py", "test_gpu_ddp_engine.py", "test_gpu_peer.py", "test_gpu_bench_two_ranks.py", "test_main_dropin.py", "test_train_ddp.py",
    "test_gpu_module.py", "test_gpu_training_equivalence.py", "test_gpu_pair_twins.py", "test_gpu_kind_kernels.py",
    "test_feature_store.py", "test_index.py", "test_gpu_avgpool.py", "test_avgpool_da.py", "test_gpu_da_extras.py",
    "test_gpu_engine_bn.py", "test_gpu_engine_mcd.py", "test_gpu_engine_avgpool_da.py", "test_gpu_da_over_ranks.py", "test_gpu_accel.py",
]


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with `-m gpu`; when the marker filter lets them through on a box without a GPU they fail
    # loudly, they do not skip.  gpu_ab tests carry no `gpu` marker: deselect them unless the -m expression names them.
    markexpr = config.getoption("-m") or ""
    if "gpu_ab" not in markexpr:
        keep, drop = [], []
        for it in items:
            (drop if it.get_closest_marker("gpu_ab") else keep).append(it)
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = keep
    rank = {name: i for i, name in enumerate(ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), len(ORDER)))     # stable: order inside a file is kept
