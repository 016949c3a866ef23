"""The compiled kernels (gfx950 ISA of the in-tree library): every workgroup barrier waits for the wave's own LDS operations.

Round 6 met a search that took tens of seconds once in a few million periods: hipcc had dropped the `s_waitcnt lgkmcnt(0)` of
__syncthreads() in front of an s_barrier whose wave still had a ds_write in flight, another wave's read behind the barrier
overtook the write and met a stale count (PERF_LOG.md, "the stalled group").  The kernels now call tlsdev::wg_sync(), which
writes the wait out; this test reads the ISA and fails on any s_barrier that is reachable, inside its basic block, from an
LDS or scalar-memory operation without that wait in between."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import REPO

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIBRARY = os.path.join(REPO, "tls_amd", "libtls_amd.so")
# (lane shuffles go through the LDS crossbar but touch no memory)
NO_MEMORY = ("ds_bpermute", "ds_permute", "ds_swizzle", "ds_nop")


def disassemble(tmp_path):
    work = tmp_path / "libtls_amd.so"
    shutil.copy(LIBRARY, work)
    subprocess.run([OBJDUMP, "--offloading", str(work)], check=True, capture_output=True, cwd=tmp_path)   # bundles land beside it
    objects = [p for p in os.listdir(tmp_path) if "gfx950" in p]
    assert len(objects) == 1, objects
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(tmp_path / objects[0])], check=True, capture_output=True, text=True)
    return out.stdout


def unguarded_barriers(text):
    """[(function, reason)] for every s_barrier without `s_waitcnt ... lgkmcnt(0)` between it and the nearest LDS / scalar-memory
    instruction (or the block's start) in front of it; also the number of barriers seen."""
    instructions = []   # ("F", name) | ("I", text)
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            instructions.append(("F", m.group(1)))
            continue
        body = line.split("//")[0].strip()
        if body:
            instructions.append(("I", body))
    found, total, function = [], 0, None
    for i, (kind, body) in enumerate(instructions):
        if kind == "F":
            function = body
            continue
        if body.split()[0] != "s_barrier":
            continue
        total += 1
        j, reason = i - 1, None
        while j >= 0:
            kind_j, body_j = instructions[j]
            if kind_j == "F":
                reason = "start of the function"
                break
            op = body_j.split()[0]
            if op == "s_waitcnt" and "lgkmcnt(0)" in body_j:
                break
            if (op.startswith("ds_") and not op.startswith(NO_MEMORY)) or op.startswith(("s_load", "s_buffer_load", "flat_")):
                reason = "behind " + body_j
                break
            if op.startswith(("s_cbranch", "s_setpc", "s_swappc")) or op in ("s_branch", "s_endpgm"):
                reason = "start of the basic block"
                break
            j -= 1
        if reason:
            found.append((function, reason))
    return found, total


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump")
def test_every_workgroup_barrier_waits_for_the_lds(tmp_path):
    found, total = unguarded_barriers(disassemble(tmp_path))
    assert total > 500, total                      # (the library holds ~1000 barriers; none seen = the disassembly failed)
    assert not found, found[:8]


def test_no_kernel_calls_syncthreads_directly():
    """The source side of the same rule: device code synchronises through wg_sync() / lds_barrier() only."""
    for name in ("tls_kernels.hip.h", "tls_search_body.inc.h", "tls_slim_kernel.hip.h", "tls_amd.hip"):
        text = open(os.path.join(REPO, "tls_amd", "csrc", name)).read()
        code = re.sub(r"//[^\n]*", "", text)
        assert "__syncthreads()" not in code, name
        assert "__builtin_amdgcn_s_barrier" not in code.replace("    __builtin_amdgcn_s_barrier();\n    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE", ""), name
