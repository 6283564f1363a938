"""Import-name shim: the reference's scripts do `import models` / `from models import ...`
(main.py:12-16, models.py:8).  Put this directory first on sys.path and they pick up
the MI355X implementation unchanged."""
from ta3n_amd.models import *  # noqa: F401,F403
from ta3n_amd.models import VideoModel, GradReverse  # noqa: F401

# the reference's loop clips and steps per tensor (main.py:578-583); with VideoModel's flat parameter / gradient buffers both are a
# few passes over flat memory, same arithmetic (ta3n_amd/accel.py; TA3N_ACCEL=0 keeps torch's own code)
from ta3n_amd import accel as _accel  # noqa: E402
_accel.install()
