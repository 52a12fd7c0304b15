"""CPU: the disk cache of run-time compiled kernels (gwinferno_amd/csrc/gwi_jit.h) is trusted only as far as it is ours.
What a cache file holds is loaded onto the GPU inside the calling process, so:
  * a cache directory must be a real directory of this user that nobody else can write -- a 0777 directory, a symbolic link or
    the predictable /tmp fallback prepared by someone else is skipped (the chain is still compiled, in this process only);
  * a cache file carries a digest of (kinds, samples per lane, kernel names, code object) and is used only if it matches --
    a bit-flipped or truncated file, or one in the previous format, is compiled over, never loaded.
hipRTC cross-compiles for gfx950 without a GPU, so all of this runs here.  Every case is a fresh process: the library also
keeps compiled chains in memory."""
import json
import os
import stat
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = [3, 6]  # PL q x PL z: the smallest chain there is (about a second to compile)

CHILD = """
import json, sys
sys.path.insert(0, %r)
from gwinferno_amd import _native as N
print(json.dumps(N.jit_compile(%r, 2)))
""" % (ROOT, KINDS)


def compile_in_a_fresh_process(cache, extra_env=None):
    env = dict(os.environ, GWI_JIT_CACHE=str(cache))
    env.pop("GWI_QUIET", None)
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1]), out.stderr


@pytest.fixture(scope="module")
def hiprtc_available(tmp_path_factory):
    d = tmp_path_factory.mktemp("probe")
    try:
        compile_in_a_fresh_process(d)
    except AssertionError as exc:  # no libhiprtc in this image: nothing to cache
        pytest.skip(f"hipRTC not usable here: {exc}")
    return True


def test_a_private_directory_is_used_and_files_are_private(hiprtc_available, tmp_path):
    cache = tmp_path / "mine"
    got, _ = compile_in_a_fresh_process(cache)
    assert got["path"].startswith(str(cache)) and not got["from_cache"] and got["compile_seconds"] > 0
    assert stat.S_IMODE(os.stat(cache).st_mode) == 0o700 and stat.S_IMODE(os.stat(got["path"]).st_mode) == 0o600
    head = open(got["path"], "rb").read(8)
    assert head == b"GWIJIT2\n"
    again, _ = compile_in_a_fresh_process(cache)
    assert again["from_cache"] and again["compile_seconds"] == 0.0 and again["path"] == got["path"]


@pytest.mark.parametrize("how", ["world_writable", "group_writable", "symlink"])
def test_a_directory_others_could_have_prepared_is_skipped(hiprtc_available, tmp_path, how):
    real = tmp_path / "real"
    real.mkdir(mode=0o700)
    if how == "symlink":
        cache = tmp_path / "link"
        os.symlink(real, cache)
    else:
        cache = real
        os.chmod(cache, 0o777 if how == "world_writable" else 0o770)
    got, err = compile_in_a_fresh_process(cache)
    assert got["path"] == "" and not got["from_cache"] and got["compile_seconds"] > 0  # compiled, in this process only
    assert os.listdir(real) == []                                                     # nothing read from it, nothing written to it
    assert "no trusted cache directory" in err and ("writable by group or others" in err or "not a directory" in err)


@pytest.mark.parametrize("damage", ["bit_flip", "truncated", "old_format", "foreign_name", "world_writable_file"])
def test_a_damaged_or_foreign_cache_file_is_compiled_over_never_loaded(hiprtc_available, tmp_path, damage):
    cache = tmp_path / "cache"
    got, _ = compile_in_a_fresh_process(cache)
    blob = bytearray(open(got["path"], "rb").read())
    elf = blob.index(b"\x7fELF")
    if damage == "bit_flip":
        blob[elf + len(blob[elf:]) // 2] ^= 0x10          # one bit somewhere in the code object; magic and ELF header intact
    elif damage == "truncated":
        del blob[elf + 4096:]
    elif damage == "old_format":
        lines = bytes(blob[:elf]).split(b"\n")             # magic, five names, digest, ''
        blob = bytearray(b"\n".join([b"GWIJIT1"] + lines[1:6]) + b"\n" + bytes(blob[elf:]))
    elif damage == "foreign_name":
        lines = bytes(blob[:elf]).split(b"\n")
        lines[1] = b"_Z6victimv; rm -rf"                   # a kernel name that is no symbol (and no longer what the digest covers)
        blob = bytearray(b"\n".join(lines) + bytes(blob[elf:]))
    if damage == "world_writable_file":
        os.chmod(got["path"], 0o666)                       # intact content, but anyone could have replaced it
    else:
        open(got["path"], "wb").write(bytes(blob))
    again, _ = compile_in_a_fresh_process(cache)
    assert not again["from_cache"] and again["compile_seconds"] > 0 and again["path"] == got["path"]
    fresh = open(got["path"], "rb").read()
    assert fresh.startswith(b"GWIJIT2\n") and stat.S_IMODE(os.stat(got["path"]).st_mode) == 0o600
    third, _ = compile_in_a_fresh_process(cache)           # ... and the rewritten file is good
    assert third["from_cache"]
