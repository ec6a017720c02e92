"""CPU test of the conv kernels' tile choice (conv_common.h pick_tile / patch_read_conflicts): the model is cut out of the
header and compiled with g++ -- no GPU, no HIP.  It pins what profiles/r2_conv_pmc.json measured: 8 x 16 tiles (or 8-wide
tiles on a padded LDS row pitch) make the patch-fragment ds_read_b128 of the stride-1 layers conflict-free."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    src = open(os.path.join(ROOT, "coivo_amd", "csrc", "conv_common.h")).read()
    a, b = src.index("struct Tile {"), src.index("int check_desc(")
    d = tmp_path_factory.mktemp("tile")
    snippet = d / "snippet.h"
    snippet.write_text(src[a:b])
    exe = d / "tile_model"
    subprocess.run(["g++", "-O2", "-std=c++17", f'-DTILE_MODEL_SNIPPET="{snippet}"', os.path.join(ROOT, "tests", "cpp", "tile_model.cpp"),
                    "-o", str(exe)], check=True)

    def query(*qs):
        args = [str(v) for q in qs for v in q]
        out = subprocess.run([str(exe)] + args, check=True, capture_output=True, text=True).stdout.split()
        vals = [float(v) for v in out]
        return [tuple(vals[i:i + 4]) for i in range(0, len(vals), 4)]
    return query


DEPTHNET = [(256, 320), (128, 160), (64, 80), (32, 40), (16, 20), (8, 10)]


def test_stride1_layers_read_without_conflicts(model):
    res = model(*[(h, w, 1, 1, 3) for h, w in DEPTHNET])
    for (h, w), (toh, tow, pwp, cf) in zip(DEPTHNET, res):
        assert toh * tow <= 128 and toh >= 1 and tow >= 1
        assert pwp >= tow + 2
        assert cf == 1.0, ((h, w), toh, tow, pwp, cf)
    # the plain cost model (what round 1 used) picks 16 x 8 on the large images: every patch read takes two passes
    plain = model((128, 160, 1, 0, 3))[0]
    assert plain[:2] == (16.0, 8.0) and plain[3] == 2.0


def test_same_tile_count_as_the_plain_model(model):
    """Conflict awareness must not cost tiles on the DepthNet shapes (it only flips the orientation or pads rows)."""
    for h, w in DEPTHNET:
        (a_toh, a_tow, _, _), (p_toh, p_tow, _, _) = model((h, w, 1, 1, 3), (h, w, 1, 0, 3))
        tiles = lambda th, tw: -(-h // int(th)) * -(-w // int(tw))
        assert tiles(a_toh, a_tow) == tiles(p_toh, p_tow), (h, w)


def test_other_patch_geometries(model):
    # k_dgrad_s2 (patch = tile + 1), k_dgrad_up2 (pixel stride 2, patch = 2 tile + 2): tiles stay within 128 positions and the
    # padded pitch never undercuts the patch width
    for q in [(64, 80, 1, 1, 2), (16, 20, 1, 1, 2), (64, 80, 2, 1, 4), (8, 10, 2, 1, 4), (1, 1, 1, 1, 3), (2, 3, 2, 1, 3)]:
        toh, tow, pwp, cf = model(q)[0]
        assert 1 <= toh * tow <= 128
        assert pwp >= (tow - 1) * q[2] + q[4]
        assert 1.0 <= cf <= 4.0
