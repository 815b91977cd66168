"""CPU: the C-ABI library loads and exports every symbol include/amp_engine.h declares (no compute calls)."""

import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "amp_engine.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(amp_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_path():
    names = declared_functions()
    for must in ("amp_motion_create", "amp_motion_sample", "amp_motion_frame_blend", "amp_collect_reference",
                 "amp_reset_reference_state", "amp_env_step", "amp_reset_compact", "amp_disc_style_reward", "amp_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from humanoid_amp_amd import _native as nat  # conftest.py rebuilt the library if it was stale

    lib = ctypes.CDLL(nat.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding covers exactly the header
    assert sorted(nat.SIGNATURES) == declared_functions()
    assert lib.amp_abi_version() == nat.ABI_VERSION


def test_no_cxx_symbols_leak_into_the_abi_names():
    """Every exported amp_* symbol is unmangled C."""
    from humanoid_amp_amd import _native as nat

    out = subprocess.run(["nm", "-D", "--defined-only", nat.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    for n in declared_functions():
        assert n in exported


def test_struct_layouts_match_the_header():
    """ctypes mirrors: spot-check sizes that would silently corrupt arguments if they drifted."""
    from humanoid_amp_amd import _native as nat

    assert ctypes.sizeof(nat.AmpMotionDesc) == 4 * 4 + 8 + 8 + 8 + 6 * 8
    assert ctypes.sizeof(nat.AmpEnvCfg) == 10 * 4 + 8 + 6 * 4 + 3 * 8
    assert ctypes.sizeof(nat.AmpSimState) == 9 * 16 + 32 + 16 + 3 * 8
    assert ctypes.sizeof(nat.AmpEnvBuffers) == 10 * 8 + 8 + 8 + 8 + 8 + 4 + 4
    assert ctypes.sizeof(nat.AmpDiscDesc) == 16 + 6 * 8


def test_errors_are_codes_not_exceptions_and_need_no_gpu():
    from humanoid_amp_amd import _native as nat

    lib = nat.load()
    assert lib.amp_env_step(None, None, None, 0, 7, None) == -1  # AMP_ERR_INVALID
    assert b"null" in lib.amp_last_error()
    assert lib.amp_reset_compact_workspace_bytes(65536) == 4 * (1024 + 1)
    assert [lib.amp_env_step_tile_envs(n) for n in (1, 4096, 32768, 65536, 1 << 20)] == [16, 16, 32, 64, 64]
    with pytest.raises(nat.AmpEngineError):
        nat.check(-1, "x")
