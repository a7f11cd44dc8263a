"""Hand-off watchdog of the one-launch dense-block kernels, at the places where results leave the device.

The fused bf16 dense block (csrc/conv_bf16.hip, ``sr_set_conv_chain(3)``, the default) makes progress only while about
six rows of its 16x32 tiles are resident; a GPU that withholds CUs from this process (shared with another process,
a partitioned device) lets its bounded waits spin out.  The kernel then raises an abort word and leaves; what it wrote
is invalid.  The C drivers copy those words to pinned host memory behind their launches and fail the NEXT driver call —
which is too late for the last forward of a process.  The reference's contract is that a forward either returns the
right image or raises (basicsr/models/sr_model.py:120-129), so every device -> host hand-over of the product goes
through this module:

``guarded(fn)``       inference: run ``fn`` (forwards only), synchronise, ask ``sr_chain_watchdog``; on a time-out log a
                      warning, switch THIS PROCESS to the chain launch (``sr_set_conv_chain(2)``: work items are
                      claimed, no residency is assumed; same bits) and run ``fn`` again.  No re-exec, no CPU path.
``verify(where)``     training / checkpoints: synchronise (optional) and raise ``SrHipError`` on a time-out — a step
                      whose gradients were invalid must not be logged or saved as if it had succeeded.
"""
import logging

import torch

from . import _lib

_TIMEOUT_MARK = 'timed out waiting for a neighbour tile'
fallback_count = 0   # how often this process fell back to the chain launch (tests, bench)


def tripped(synchronize=True):
    """True when a dense-block launch issued before this call timed out (the record is cleared)."""
    if synchronize and torch.cuda.is_available():
        torch.cuda.synchronize()
    return _lib.load().sr_chain_watchdog() != 0


def is_timeout(exc):
    return isinstance(exc, _lib.SrHipError) and _TIMEOUT_MARK in str(exc)


def verify(where, synchronize=True):
    """Raises when a dense-block launch timed out.  ``synchronize=False`` when the caller has just synchronised
    (a ``.item()`` / ``.tolist()`` / ``.cpu()`` on the same stream)."""
    lib = _lib.load()
    if synchronize and torch.cuda.is_available():
        torch.cuda.synchronize()
    rc = lib.sr_chain_watchdog()
    if rc != 0:
        raise _lib.SrHipError(f'{where}: {lib.sr_last_error().decode("utf-8", "replace")}')


def guarded(fn, what='forward'):
    """``fn()`` with the hand-off watchdog: its result is only returned when no dense-block launch timed out.
    ``fn`` must be repeatable (forwards without side effects)."""
    global fallback_count
    try:
        out = fn()
        bad = tripped()
    except _lib.SrHipError as exc:     # a later driver call of fn() itself found the earlier time-out
        if not is_timeout(exc):
            raise
        tripped()                      # drain what is still in flight, clear the record
        bad = True
    if not bad:
        return out
    fallback_count += 1
    logging.getLogger('basicsr').warning(
        f'{what}: a fused dense-block launch timed out waiting for neighbour tiles (is the GPU shared?); its results '
        'are discarded and the work is repeated on the chain launch (sr_set_conv_chain(2)), which this process keeps '
        'using from now on.')
    _lib.check(_lib.load().sr_set_conv_chain(2), 'sr_set_conv_chain')
    out = fn()
    verify(f'{what} (repeated on the chain launch)')
    return out
