"""Test switches for the scripts under tools/: `set(name, value)` goes through the C ABI's test hook (pilot_ot_test_switch); the
library itself reads no environment variable.  Shell drivers pass switches to a tool as TOOL_SWITCHES="NAME=VALUE,NAME=VALUE",
which `apply_from_env()` (called by the tools that are driven that way) turns into hook calls."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def set(name, value):
    from pilot_amd import _lib
    _lib.test_switch(name, None if value is None else str(value))


def apply_from_env():
    for item in filter(None, os.environ.get("TOOL_SWITCHES", "").split(",")):
        name, _, value = item.partition("=")
        set(name.strip(), value.strip())
