"""k_schur_syrk4 (v_mfma_f64_4x4x4_4b register tiles, LDS-DMA staging, three-buffer pipeline) against the plain
16x16x4 kernel k_schur_syrk on random Z: the partial sums of every chunk and block must be the same bit for bit (same
products, same order), for full and ragged chunks, widths from one column group to the kernel's limit, and chunks
shorter than a stage.  Runs scripts/micro/syrk4_bench (built by cvx_amd/csrc/Makefile); the harness exits non-zero on
any mismatch."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "scripts", "micro", "syrk4_bench")

SHAPES = [
    (336, 60000, 128),   # BASELINE config 4: 21 column groups, 128 chunks (ragged: 468.75 rows each)
    (336, 61440, 128),   # the same with whole stages only (480 rows per chunk)
    (400, 30000, 86),    # the widest the kernel takes: 25 groups, three parts, the 128-byte fourth piece of a row
    (320, 30000, 128),   # 20 groups: no edge group riding in the DIAG tasks
    (64, 3000, 128),     # one task per kind, chunks of 23 rows
    (144, 3001, 7),      # border tiles (GEN bodies), odd row count
    (336, 40, 8),        # chunks of five rows: less than a stage
    (80, 10, 1),         # a single short chunk
    (16, 100, 3),        # one column group
]


@pytest.mark.gpu
@pytest.mark.parametrize("zc,rows,chunks", SHAPES)
def test_register_tile_kernel_equals_the_16x16x4_kernel_bit_for_bit(zc, rows, chunks):
    assert os.path.exists(EXE), "run __graft_entry__.build() (make -C cvx_amd/csrc)"
    p = subprocess.run([EXE, str(zc), str(rows), str(chunks)], capture_output=True, timeout=110)
    out = p.stdout.decode()
    m = re.search(r"max \|diff\| (\S+) .*mismatches (\d+) of (\d+)", out)
    assert m, out + p.stderr.decode()
    assert p.returncode == 0 and int(m.group(2)) == 0 and float(m.group(1)) == 0.0, out
    assert int(m.group(3)) == chunks * (zc // 16) * (zc // 16 + 1) // 2 * 256
