"""Test hooks that steer the kernels onto their rarely taken paths (heavy-tile split, pool-slot overflow, the
alternative lMHL layouts).  None of them changes a result: every setting must give the oracle's table."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("env", [
    {},                                                                  # defaults
    {"EPIHIP_CX_ABLATE": "7", "EPIHIP_MHL_ABLATE": "4095", "EPIHIP_CX_DIAG": "1", "EPIHIP_CX_PACKED": "0"},   # round-1 timing switches: the
                                                                         # product library no longer reads them (timing builds are a make target)
    {"EPIHIP_HEAVY_ROWS": "500"},                                        # pile-ups split over many workgroups (slabs in HBM)
    {"EPIHIP_HEAVY_ROWS": "100"},                                        # ... in chunks smaller than the u8 fold interval
    {"EPIHIP_CX_SLOT": "3", "EPIHIP_HEAVY_ROWS": "500"},                 # nearly every tile outgrows its pool slot
    {"EPIHIP_PR_WIDE": "0"},                                             # per-read kernels: the 2-lanes-per-read layout for every call
    {"EPIHIP_CX_SLOT": "0"},                                             # no slots: every tile through the cursor
    {"EPIHIP_TILE_HINT": "0"},                                           # tile index counted and scanned by every call (no remembered offsets)
    {"EPIHIP_REALIGN": "0"},                                             # rows stay back to back (no position-congruent copy, layout.hip)
    {"EPIHIP_REALIGN": "4"},                                             # ... congruent to their start position modulo 4 only
    {"EPIHIP_REALIGN": "0", "EPIHIP_MHL_FUSED": "0", "EPIHIP_CX_LEAN": "0"},
    {"EPIHIP_CX_LEAN": "0"},                                             # single-context CX reports: the general kernel (u16 copy of the
                                                                         # u8 counters, folds) also where no position is deeper than 255 rows
    {"EPIHIP_CX_LEAN": "0", "EPIHIP_HEAVY_ROWS": "100", "EPIHIP_CX_SLOT": "3"},
    {"EPIHIP_MHL_SLOT": "2", "EPIHIP_HEAVY_ROWS": "500"},                # lMHL (fused kernel): nearly every tile outgrows its pool slot
    {"EPIHIP_MHL_SLOT": "0"},
    {"EPIHIP_HEAVY_ROWS": "40"},                                         # fused lMHL kernel gives up on tiles with > 40 rows: two-kernel path
    {"EPIHIP_MHLF_FOLD": "0"},                                           # one-pass lMHL kernel without the LDS fold array: tiles over 255 rows
                                                                         # fold their call counters into slab slots in HBM
    {"EPIHIP_MHLF_FOLD": "0", "EPIHIP_MHLF_FOLD_SLOTS": "1"},            # ... and find no slot left: the WIDE variant takes them
    {"EPIHIP_MHLF_FOLD": "1"},                                           # ... with it, whatever the batch looks like
    {"EPIHIP_MHL_FUSED": "0"},                                           # lMHL: the two-kernel path (records) for every batch
    {"EPIHIP_MHL_FUSED": "0", "EPIHIP_MHL_SLOT": "2", "EPIHIP_HEAVY_ROWS": "500"},
    {"EPIHIP_MHL_FUSED": "0", "EPIHIP_MHL_SLOT": "0"},
    {"EPIHIP_MHL_FUSED": "0", "EPIHIP_MHL_WG": "512", "EPIHIP_HEAVY_ROWS": "500"},   # lMHL tile kernel with 512-thread workgroups for short reads too
    {"EPIHIP_MHL_WG": "256", "EPIHIP_MHL_MULTI": "1"},                   # ... and 256 with the per-block records of long reads
    {"EPIHIP_MHL_MULTI": "1"},                                           # lMHL pass 1: wavefront-per-read kernel for every read
    {"EPIHIP_MHL_MULTI": "1", "EPIHIP_MHL_TILE_GROUP": "64", "EPIHIP_HEAVY_ROWS": "500"},
    {"EPIHIP_MHL_FUSED": "0", "EPIHIP_MHL_SUMS": "64", "EPIHIP_HEAVY_ROWS": "500"},   # lMHL pass 2 with u64 LDS sums where u32 would do
])
def test_cx_kernel_variants(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_variant_worker.py")], env=e, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "variant ok" in r.stdout
