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


def test_struct_layouts_match_the_header(tmp_path):
    """ctypes mirrors vs the header itself: gcc compiles include/amp_engine.h as C and reports sizeof / offsetof of
    every struct and field the binding passes by pointer; a drift would silently corrupt arguments."""
    import subprocess

    from humanoid_amp_amd import _native as nat

    structs = {"AmpMotionDesc": nat.AmpMotionDesc, "AmpEnvCfg": nat.AmpEnvCfg, "AmpSimState": nat.AmpSimState,
               "AmpEnvBuffers": nat.AmpEnvBuffers, "AmpDiscDesc": nat.AmpDiscDesc, "AmpDiscInputLayout": nat.AmpDiscInputLayout, "AmpDiscPlanInfo": nat.AmpDiscPlanInfo,
               "AmpResetArgs": nat.AmpResetArgs, "AmpCommandArgs": nat.AmpCommandArgs, "AmpCompactArgs": nat.AmpCompactArgs, "AmpScatterRows": nat.AmpScatterRows, "AmpRewardLogArgs": nat.AmpRewardLogArgs, "AmpPrePhysicsArgs": nat.AmpPrePhysicsArgs, "AmpHotStepArgs": nat.AmpHotStepArgs, "AmpDiscTrainCfg": nat.AmpDiscTrainCfg, "AmpKinModel": nat.AmpKinModel,
               "AmpConvertOutputs": nat.AmpConvertOutputs}
    lines = ["#include <stddef.h>", "#include <stdio.h>", '#include "amp_engine.h"', "int main(void) {"]
    for name, cls in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf("{name}.{field} %zu\\n", offsetof({name}, {field}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == ctypes.sizeof(cls), name
        for field, _ in cls._fields_:
            assert int(got[f"{name}.{field}"]) == getattr(cls, field).offset, f"{name}.{field}"


def test_errors_are_codes_not_exceptions_and_need_no_gpu():
    from humanoid_amp_amd import _native as nat

    lib = nat.load()
    assert lib.amp_env_step(None, None, None, 0, 7, None) == -1  # AMP_ERR_INVALID
    assert b"null" in lib.amp_last_error()
    assert lib.amp_reset_compact_workspace_bytes(65536) == 4 * (1024 + 1)
    cfg = nat.AmpEnvCfg(n_dof=29, n_key=4, num_amp_observations=2, num_actor_observations=1)
    assert [lib.amp_env_step_tile_envs(ctypes.byref(cfg), n) for n in (1, 4096, 16383, 16384, 65536, 1 << 20)] == [16, 16, 16, 32, 32, 32]
    cfg.num_amp_observations = 10  # the [tile, K*D] LDS image of the DMA body bounds the tile
    assert [lib.amp_env_step_tile_envs(ctypes.byref(cfg), n) for n in (4096, 65536)] == [8, 8]
    assert lib.amp_env_step_tile_envs(None, 4096) == -1
    with pytest.raises(nat.AmpEngineError):
        nat.check(-1, "x")
