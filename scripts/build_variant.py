#!/usr/bin/env python3
"""Build an A/B variant of the library: scripts/build_variant.py <name> [-DFLAG=..] ...  -> gym_dockauv_amd/lib/libdockauv_<name>.so
(select it at run time with DOCKAUV_LIB=<path>)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name, flags = sys.argv[1], sys.argv[2:]
os.environ["DOCKAUV_LIB_NAME"] = f"libdockauv_{name}.so"
os.environ["DOCKAUV_OBJ_TAG"] = "_" + name
from gym_dockauv_amd.csrc import build  # noqa: E402
print(build.build(force=True, extra=flags))
