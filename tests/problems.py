"""The seeded synthetic generators live in the package (clarabel.rs_amd/synthetic.py) so that
bench.py and the tools never import `tests`; the tests keep this name as an alias."""
import sys

import __graft_entry__ as _g

_g.load_package()
import clarabel_rs_amd.synthetic as _syn  # noqa: E402

sys.modules[__name__] = _syn
