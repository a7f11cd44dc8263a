"""The static schedule of the fused dense-block kernel (csrc/fused_sched.h) against an independent model.

The kernel never drains its memory queue: every step begins with ``s_waitcnt vmcnt(K)`` + barrier, K computed at compile time by
``make_sched``.  Here the schedule is compiled with g++ (it is plain constexpr C++), printed, and REPLAYED in Python with a model written
from the kernel's issue order only: a list of vector-memory operations per wave in program order, in-order retirement (``vmcnt(K)`` =
all but the K youngest are complete).  Checked for every epilogue mode, for the first tile of a workgroup and for a following one
(whose x tile and first weight groups were issued during the previous tile):
  * what a step's MFMAs read has landed at its barrier — its weight pieces and tile chunk, and the operands that the PREVIOUS step
    reads ahead for it (tile chunk, first AR-1 weight fragments);
  * no LDS-DMA overwrites a ring slot or a tile buffer whose content is still read (read-ahead included);
  * a conv's stores are covered by the wait of the step that publishes it; flag fetch and ticket atomic by their own waits;
  * every accumulator group is zeroed exactly when it first appears; the per-accumulator MFMA order is chunk-major, then tap column,
    then tap row (the order of the conv-by-conv kernel: bit-identical results)."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, '..', 'image_restoration_amd', 'csrc')


# the three instances conv_bf16.hip builds: 16-row tiles (the header's defaults), 8-row tiles with a chunk per step, and the 16-row
# tile on four waves of four rows (the macros conv_bf16.hip sets in front of its second / third inclusion of fused_block.inc)
INSTANCES = {
    'rows16': [],
    'rows16_waves4': ['-DSR_FZ_NW=4', '-DSR_FZ_PT=4'],
    'rows8': ['-DSR_FZ_PT=1', '-DSR_FZ_RING=64', '-DSR_FZ_PERDX=0', '-DSR_FZ_CLAIMLEAD=2', '-DSR_FZ_PUBLAG={1,1,1,1}', '-DSR_FZ_TILELAG={2,2,2,2}',
              '-DSR_FZ_FLAGLEAD={1,1,1,1}'],
}


@pytest.fixture(scope='module', params=sorted(INSTANCES))
def schedules(request, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('sched') / ('print_fused_sched_' + request.param))
    subprocess.run(['g++', '-std=c++17', '-O1', '-fconstexpr-ops-limit=2000000000', '-fconstexpr-loop-limit=100000000', '-I', CSRC]
                   + INSTANCES[request.param] + [os.path.join(HERE, 'helpers', 'print_fused_sched.cpp'), '-o', exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    return [json.loads(line) for line in out.strip().splitlines()]


IN_TB0 = {0: 0, 1: 4, 2: 0, 3: 2, 4: 4}
IN_CHUNKS = {0: 4, 1: 2, 2: 2, 3: 2, 4: 2}


def replay(s, following, columns_wave=False):
    """Issue sequence of one wave for one tile.  Returns (ops, per-step bookkeeping).  An op = (kind, payload, issue step) with step -1 =
    before the tile's first step (prologue or, for a following tile, the previous tile's read-ahead)."""
    mode, st, ring, ar = s['mode'], s['steps'], s['ring'], s['ar']
    ops = []

    def issue(kind, payload, step):
        # two tile chunks = 2 XU pieces of 1 KB, TPW per wave (40 = five each on 16-row tiles with eight waves); a weight group = eight
        # pieces, GPW per wave
        for _ in range(s['tpw'] if kind in ('tile', 'nxtile') else s['gpw'] if kind in ('w', 'nxw') else 1):
            ops.append((kind, payload, step))
        return len(ops)          # position: number of ops issued up to and including this one

    pos = {}                     # ('w', group) / ('t', in, pair) / ('st', conv) / ('flag', in) / ('claim',) -> position
    # what precedes the tile: x pairs, the first q_ahead groups — in a following tile they were issued during the previous tile,
    # interleaved with that tile's last operations (which are older than everything below or in between: modelled as extra ops)
    if following:
        prev = s['steps']
        for i, d in enumerate(prev):
            if d['nx_tile']:
                pos[('t', 0, d['nx_tile'] - 1)] = issue('tile', (0, d['nx_tile'] - 1), -1)
            for q in range(d['nx_q0'], d['nx_q1']):
                pos[('w', q)] = issue('w', q, -1)
            if d['nx_tile'] or d['nx_q1'] > d['nx_q0']:
                issue('other', 'previous tile', -1)   # e.g. its remaining weight groups, residual loads, stores: younger than some of the above
    else:
        pos[('t', 0, 0)] = issue('tile', (0, 0), -1)
        pos[('t', 0, 1)] = issue('tile', (0, 1), -1)
        for q in range(s['q_ahead']):
            pos[('w', q)] = issue('w', q, -1)
    for q in range(s['q_ahead'], s['q_init']):
        pos[('w', q)] = issue('w', q, -1)
    waits = []
    for i, d in enumerate(st):
        waits.append(len(ops))   # ops issued when the step's wait executes
        if d['claim'] == 2:
            pos[('hand',)] = len(ops)
        if d['tile_in']:
            if mode != 0 and s['halo']:
                # lean instances: only the ring of an x1..x4 tile comes through memory — HPW in-place pieces per wave, and waves 0 / 1
                # issue the staging piece of the halo columns right behind theirs
                for _ in range(s['hpw'] + (1 if columns_wave else 0)):
                    pos[('t', d['tile_in'], 0)] = issue('halo', (d['tile_in'], 0), i)
            else:
                pos[('t', d['tile_in'], 0)] = issue('tile', (d['tile_in'], 0), i)
        for q in range(d['q0'], d['q1']):
            pos[('w', q)] = issue('w', q, i)
        if d['nx_tile']:
            issue('nxtile', d['nx_tile'] - 1, i)
        for q in range(d['nx_q0'], d['nx_q1']):
            issue('nxw', q, i)
        if d['flag_in']:
            pos[('flagfetch', d['flag_in'])] = len(ops)     # wave 0 only: its extra op sits here, the others' waits only get stricter
        if mode != 0 and d['first_of_in'] == 4:
            for _ in range(8 * s['pt']):   # 2 sources x 2 cout tiles x PT rows x 2 channel blocks
                issue('res', None, i)
        if mode == 2 and d['mask_conv']:
            for _ in range(2 * s['pt']):
                issue('mask', d['mask_conv'], i)
        if d['claim'] == 1:
            pos[('claim',)] = len(ops)                      # thread 0 only
        if 1 <= d['post'] <= 4:
            for _ in range(2 * s['pt']):   # PT rows x 2 channel blocks
                pos[('st', d['post'])] = issue('store', d['post'], i)
    return ops, pos, waits


@pytest.mark.parametrize('columns_wave', [False, True])
@pytest.mark.parametrize('following', [False, True])
def test_counted_waits_cover_what_each_step_reads(schedules, following, columns_wave):
    for s in schedules:
        assert s['ok'] == 1 and s['npieces'] == 468 and s['nsteps'] <= 80
        st, ar = s['steps'], s['ar']
        ops, pos, waits = replay(s, following, columns_wave)
        for i, d in enumerate(st):
            done = waits[i] - d['K']                       # ops with position <= done are complete behind this step's wait
            need = []
            for j in (i, i + 1):
                if j >= len(st):
                    continue
                dj = st[j]
                last_piece = dj['wp0'] + dj['wpn'] - 1 if j == i else dj['wp0'] + ar - 2
                need.append(('w', last_piece // 8))
                need.append(('t', dj['in'], dj['chunk'] // 2))
            if d['publish']:
                need.append(('st', d['publish']))
            for key in need:
                assert key in pos, (s['mode'], i, key, 'never issued before it is needed')
                assert pos[key] <= done, (s['mode'], i, key, pos[key], done)
            assert 0 <= d['K'] <= 60
            if d['tile_in']:                               # wave 0: the flag fetch of this input is covered by its own wait
                assert waits[i] - d['Kflag'] >= pos[('flagfetch', d['tile_in'])]
            if d['claim'] == 2:                            # thread 0: the ticket atomic
                assert pos[('hand',)] - d['Kclaim'] >= pos[('claim',)]
            if d['copy_in']:                               # the halo columns are copied out of staging behind this step's barrier: landed
                assert pos[('t', d['copy_in'], 0)] <= done, (s['mode'], i, 'staging not landed at the copy')
                assert st[i + 1]['first_of_in'] == d['copy_in'] and st[i + 1]['pre'] == 0, 'the first use must not be read ahead of the copy'


def test_ring_slots_and_tile_buffers_are_not_overwritten_while_read(schedules):
    for s in schedules:
        st, ring, ar = s['steps'], s['ring'], s['ar']
        # step at which each piece is last read: its own step (the read-ahead of the step before reads it EARLIER, never later)
        read_step = {}
        for i, d in enumerate(st):
            for p in range(d['wp0'], d['wp0'] + d['wpn']):
                read_step[p] = i
        issue_step = {}
        for q in range(s['q_init']):
            for p in range(8 * q, 8 * q + 8):
                issue_step[p] = -1
        for i, d in enumerate(st):
            for q in range(d['q0'], d['q1']):
                for p in range(8 * q, 8 * q + 8):
                    issue_step[p] = i
        for p, i in issue_step.items():
            old = p - ring
            if old >= 0 and old in read_step:
                # a DMA issued behind the barrier of step i may only replace what steps < i read
                assert read_step[old] < i, (s['mode'], p, old, read_step[old], i)
            if p in read_step:
                assert i < read_step[p] or i == -1, (s['mode'], p, 'issued too late')
        # next tile's first groups: their slots' last occupants are consumed
        total = s['ngroups'] * 8
        for i, d in enumerate(st):
            for q in range(d['nx_q0'], d['nx_q1']):
                for j in range(8 * q, 8 * q + 8):
                    occ = j + ring * ((total - 1 - j) // ring)
                    if occ in read_step:
                        assert read_step[occ] < i, (s['mode'], q, occ)
        # tile buffers: a pair is written by the tile of input t at step ti; everything read from those buffers at steps >= ti belongs to t
        first_use = {}
        last_use = {}
        for i, d in enumerate(st):
            first_use.setdefault(d['in'], i)
            last_use[d['in']] = i
        for i, d in enumerate(st):
            if d['tile_in']:
                t = d['tile_in']
                bufs = {IN_TB0[t], IN_TB0[t] + 1}
                assert i < first_use[t] - 1, (s['mode'], t, 'the read-ahead of the step before the first use needs the tile')
                for j, dj in enumerate(st):
                    if dj['tb'] in bufs and j >= i - 0:
                        assert dj['in'] == t or j > last_use[t], (s['mode'], t, j)
                    if dj['tb'] in bufs and dj['in'] != t and j <= last_use[t]:
                        assert j < i, (s['mode'], t, j, 'another input is read from these buffers while t owns them')
            if d['copy_in'] and s['halo']:
                # one staging KB per chunk serves x1..x4 in turn: the next input's halo columns are issued only after this copy
                later = [j for j, dj in enumerate(st) if dj['tile_in'] == d['copy_in'] + 1]
                assert all(j > i for j in later), (s['mode'], d['copy_in'], later, i)
            if 1 <= d['post'] <= 4 and s['halo']:
                # lean instances: conv t's epilogue writes its tile into input t's buffers at this step — whoever else used them is done,
                # and input t's own readers come later, behind a barrier
                t = d['post']
                bufs = {IN_TB0[t], IN_TB0[t] + 1}
                for j, dj in enumerate(st):
                    if dj['tb'] in bufs and dj['in'] != t:
                        assert j < i or j > last_use[t], (s['mode'], t, j, 'still read when the epilogue overwrites it')
                    if dj['tb'] in bufs and dj['in'] == t:
                        assert j > i + 1, (s['mode'], t, j)
            if d['nx_tile']:
                bufs = {2 * (d['nx_tile'] - 1), 2 * (d['nx_tile'] - 1) + 1}
                assert all(dj['tb'] not in bufs for dj in st[i:]), (s['mode'], i, 'next tile lands on data still read')


def test_accumulators_and_mfma_order(schedules):
    s = schedules[0]
    seen, order = set(), {g: [] for g in range(6)}
    for d in s['steps']:
        new = {g for g in range(6) if d['zero_mask'] >> g & 1}
        groups = set(range(d['g0'], d['g0'] + d['ng']))
        assert new == groups - seen
        seen |= groups
        cb = (0 if d['in'] == 0 else 4 + 2 * (d['in'] - 1)) + d['chunk']
        for dx in range(d['dx0'], d['dx0'] + d['ndx']):
            for g in groups:
                order[g].append((cb, dx))
    for g in range(6):
        conv = min(g, 4)
        expect = [(cb, dx) for cb in range(4 + 2 * conv) for dx in range(3)]
        assert order[g] == expect, g
    posts = [d['post'] for d in s['steps'] if d['post']]
    assert posts == [1, 2, 3, 4, 5]


def test_model_instances_are_the_ones_the_library_builds():
    """INSTANCES['rows8'] above must be the macro set conv_bf16.hip puts in front of its second inclusion of fused_block.inc."""
    import re
    src = open(os.path.join(CSRC, 'conv_bf16.hip')).read()
    block = src[src.index('#define SR_FZ_NS fz8'):]
    block = block[:block.index('#include "fused_block.inc"')]
    defs = dict(re.findall(r'#define (SR_FZ_\w+) (.+)', block))
    want = {k: v.replace(' ', '') for k, v in defs.items() if k not in ('SR_FZ_NS', 'SR_FZ_KERNEL')}
    have = dict(m.lstrip('-D').split('=', 1) for m in INSTANCES['rows8'])
    assert want == have, (want, have)
