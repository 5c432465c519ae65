"""`arrow_gpu::kernels` — re-export of every op family (crates/arrow/src/kernels.rs:1-8).
Importing this package attaches the trait methods (add, eq, sin, cast, take, …) to the array classes."""
from ..array import broadcast_dyn, broadcast_op_dyn  # noqa: F401
from .arithmetic import *  # noqa: F401,F403
from .cast import *  # noqa: F401,F403
from .compare import *  # noqa: F401,F403
from .fused import FusedChain  # noqa: F401
from .logical import *  # noqa: F401,F403
from .math import *  # noqa: F401,F403
from .routines import *  # noqa: F401,F403
from .trigonometry import *  # noqa: F401,F403
