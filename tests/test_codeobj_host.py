"""What the compiler made of the kernels whose design depends on it (runs without a GPU: reads libsr_hip.so's gfx950 code objects).

* ``rdb_fused_bf16_kernel<0..2>``, ``conv_stream_bf16_kernel<...>`` and the bf16 weight-gradient kernels (``wgrad_bf16_kernel<...>``,
  ``wgrad_rdb_bf16_kernel<2|4>``) are one workgroup of 8 waves per CU with 150-159 KB of LDS and
  counted ``s_waitcnt vmcnt`` schedules computed from the issue order of a wave: a register spill adds ``scratch_load`` / ``scratch_store``
  to that order (they share ``vmcnt`` with the LDS-DMA) and the design collapses silently — slower, and in the fused kernel the
  counted waits would cover the wrong operations.  So: no scratch, at most 256 VGPRs, no spills.
* The fused kernel's tile ticket is a returning ``global_atomic_add`` issued as inline asm and consumed three steps later behind a
  counted wait.  The compiler believes the destination register is defined right after the asm; nothing may touch that register
  between the atomic and the first ``s_waitcnt vmcnt`` that follows its issue... precisely: the first instruction that reads or
  writes it must come after an ``s_waitcnt vmcnt`` (checked on the disassembly)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'image_restoration_amd', 'lib', 'libsr_hip.so')
LLVM = '/opt/rocm/lib/llvm/bin'


@pytest.fixture(scope='module')
def code_objects(tmp_path_factory):
    """[(path, kernel name -> metadata dict)] of every gfx950 code object bundled in the library."""
    for tool in ('llvm-objdump', 'llvm-readelf'):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.fail(f'{tool} is missing from {LLVM}')
    work = tmp_path_factory.mktemp('codeobj')
    lib = shutil.copy(LIB, work / 'libsr_hip.so')     # llvm-objdump --offloading extracts next to its input
    subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', lib], check=True, capture_output=True, cwd=work)
    out = []
    for f in sorted(os.listdir(work)):
        if 'gfx950' not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', str(work / f)], check=True, capture_output=True,
                               text=True).stdout
        kernels, cur = {}, None
        for line in notes.splitlines():
            m = re.match(r'\s+(?:- )?\.(\w+):\s+(\S+)', line)
            if not m:
                continue
            key, val = m.groups()
            if key == 'name' and val.startswith('_Z'):
                cur = kernels.setdefault(val, {})
            elif cur is not None and key in ('vgpr_count', 'sgpr_count', 'private_segment_fixed_size', 'vgpr_spill_count',
                                             'sgpr_spill_count', 'group_segment_fixed_size', 'agpr_count'):
                cur[key] = int(val)
        out.append((str(work / f), kernels))
    assert out, 'no gfx950 code object in libsr_hip.so'
    return out


def _kernels(code_objects, needle):
    return {n: (path, md) for path, ks in code_objects for n, md in ks.items() if needle in n}


# (round 4: wgrad_f32_rdb_kernel switches between five instances of the fp32 weight-gradient body — 144 accumulator registers each —
# inside one kernel: a spill there would put scratch traffic into every dense block's backward)
# (rdb_fused4_bf16_kernel: the 16-row tile on four waves, one per SIMD — 512 registers per lane, accumulators in the AGPR half)
@pytest.mark.parametrize('needle,at_least', [('rdb_fused_bf16_kernel', 3), ('rdb_fused8_bf16_kernel', 3), ('rdb_fused4_bf16_kernel', 2), ('conv_stream_bf16_kernel', 4),
                                             ('wgrad_bf16_kernel', 10), ('wgrad_rdb_bf16_kernel', 2), ('wgrad_f32_rdb_kernel', 1), ('wgrad_f32_kernel', 5)])
def test_one_workgroup_per_cu_kernels_have_no_scratch_and_no_spills(code_objects, needle, at_least):
    ks = _kernels(code_objects, needle)
    assert len(ks) >= at_least, sorted(ks)
    for name, (_, md) in ks.items():
        assert md['private_segment_fixed_size'] == 0, (name, md)
        assert md['vgpr_spill_count'] == 0, (name, md)   # (SGPR spills go to VGPR lanes, not to memory: not a concern here)
        assert md['vgpr_count'] <= (512 if 'rdb_fused4' in name else 256), (name, md)


def _regs(operands):
    """VGPR numbers named by an operand string (v12, v[8:11])."""
    regs = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', operands):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(r) for r in re.findall(r'\bv(\d+)\b', operands))
    return regs


@pytest.mark.parametrize('needle,instances', [('rdb_fused_bf16_kernel', 3), ('rdb_fused8_bf16_kernel', 3), ('rdb_fused4_bf16_kernel', 2)])   # 16-row, 8-row, 16-row on four waves
def test_fused_kernel_ticket_register_is_untouched_until_its_counted_wait(code_objects, needle, instances):
    ks = _kernels(code_objects, needle)
    assert len(ks) == instances
    path = next(iter(ks.values()))[0]
    dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', path], check=True, capture_output=True,
                         text=True).stdout.splitlines()
    checked = 0
    for name in ks:
        start = next(i for i, l in enumerate(dis) if l.rstrip().endswith(f'<{name}>:'))
        end = next((i for i in range(start + 1, len(dis)) if re.match(r'^[0-9a-f]+ <', dis[i])), len(dis))
        body = [l.split('//')[0].strip() for l in dis[start + 1:end]]
        atomics = [i for i, l in enumerate(body) if l.startswith('global_atomic_add ') and 'sc0' in l]
        assert len(atomics) == 2, (name, len(atomics))   # the first claim (builtin) and the in-round ticket (inline asm)
        for at in atomics:
            dst = int(re.match(r'global_atomic_add v(\d+),', body[at]).group(1))
            waited = False
            for l in body[at + 1:]:
                if l.startswith('s_waitcnt') and 'vmcnt' in l:
                    waited = True
                    continue
                op, _, operands = l.partition(' ')
                if dst in _regs(operands):
                    assert waited, f'{name}: "{l}" touches v{dst} before any vmcnt wait behind "{body[at]}"'
                    checked += 1
                    break
            else:
                pytest.fail(f'{name}: the ticket in v{dst} is never consumed')
    assert checked == 2 * instances


# (the 16-row instances: their schedules have no step whose counted wait is 0; the 8-row instance's short lags do)
@pytest.mark.parametrize('needle,instances', [('rdb_fused_bf16_kernel', 3), ('rdb_fused4_bf16_kernel', 2)])
def test_fused_kernel_never_drains_its_memory_queue_inside_a_tile(code_objects, needle, instances):
    """Every wait of the step stream is a COUNTED one (fused_sched.h).  Round 4 found an ``s_waitcnt vmcnt(0)`` that hipcc had put behind
    the barrier of each of the four steps that inspect neighbour flags (a compiler-visible store / load in the stamp and slow-poll code
    whose registers were reused right after): a full drain of the LDS-DMA read-ahead, for every wave, stamps on or off.  Those
    accesses are inline asm now; what may remain are the drains of the slow path itself (directly behind its agent-scope load / store),
    of the kernel's prologue (before the first MFMA) and the one at the kernel's end.  The lean instances (forward and transposed block);
    the generic-epilogue instance <0> loads bias / residual / mask values inside its epilogues and waits for them."""
    ks = _kernels(code_objects, needle)
    assert len(ks) == instances
    path = next(iter(ks.values()))[0]
    dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', path], check=True, capture_output=True,
                         text=True).stdout.splitlines()
    lean = [n for n in ks if 'ILi0E' not in n]
    assert len(lean) == 2
    for name in lean:
        start = next(i for i, l in enumerate(dis) if l.rstrip().endswith(f'<{name}>:'))
        end = next((i for i in range(start + 1, len(dis)) if re.match(r'^[0-9a-f]+ <', dis[i])), len(dis))
        body = [l.split('//')[0].strip() for l in dis[start + 1:end]]
        body = [l for l in body if l]
        first_mfma = min(i for i, l in enumerate(body) if l.startswith('v_mfma'))
        last_mfma = max(i for i, l in enumerate(body) if l.startswith('v_mfma'))
        stray = []
        for i, l in enumerate(body):
            if l.startswith('s_waitcnt') and 'vmcnt(0)' in l and first_mfma < i < last_mfma:
                prev = body[i - 1]
                if not ((prev.startswith('global_load_dword ') or prev.startswith('global_store_dword ')) and 'sc1' in prev):
                    stray.append((i, prev, l))
        # (the eight-wave transposed block fetches conv2's mask in the step that needs it — no register to hold it longer — and waits)
        allowed = 1 if needle == 'rdb_fused_bf16_kernel' and 'ILi2E' in name else 0
        assert len(stray) == allowed, (name, stray[:4])
