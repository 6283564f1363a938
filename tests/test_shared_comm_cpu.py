"""Host-side wiring that only a GPU run would otherwise reach: TrainEngine(share_comm=True) asks ta3n_amd.parallel for shared_native_comm
(round 6: a missing attribute there fell back to torch.distributed silently on the GPU box - the fallback worked, the intent did not)."""
import inspect

from ta3n_amd import engine, parallel


def test_shared_native_comm_exists_and_is_what_the_engine_calls():
    assert callable(parallel.shared_native_comm) and parallel.NativeComm.shared is False
    src = inspect.getsource(engine.TrainEngine.__init__)
    assert "parallel.shared_native_comm if share_comm else parallel.NativeComm" in src
    assert "share_comm" in inspect.signature(engine.TrainEngine.__init__).parameters
