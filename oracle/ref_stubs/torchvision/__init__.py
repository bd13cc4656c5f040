"""Stand-in so that `/root/reference/src/utility/functions.py` imports (torchvision is absent offline).
TEST INFRASTRUCTURE (oracle/gen_golden.py only).  No arithmetic lives here: image decoding is NOT pinned by it."""
from . import transforms  # noqa: F401
