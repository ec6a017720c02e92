// Host-only harness around conv_common.h's tile model (pick_tile / patch_read_conflicts), built with g++ by
// tests/test_tile_model_cpu.py.  Prints one line per query: "toh tow pwp conflicts".
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

#include "../../coivo_amd/csrc/tuning.h"      // the thresholds the model reads (plain C++)
using namespace colvo;

#include TILE_MODEL_SNIPPET      // the Tile struct, patch_read_conflicts() and pick_tile(), cut out of conv_common.h by the test

int main(int argc, char** argv) {
    for (int i = 1; i + 5 < argc + 1 && i + 4 < argc; i += 5) {
        const int Ho = atoi(argv[i]), Wo = atoi(argv[i + 1]), stride = atoi(argv[i + 2]), aware = atoi(argv[i + 3]), ext = atoi(argv[i + 4]);
        const Tile t = pick_tile(Ho, Wo, stride, false, 128, aware != 0, ext);
        const int pw = (t.tow - 1) * stride + ext;
        const int pwp = t.pwp > pw ? t.pwp : pw;
        printf("%d %d %d %.4f\n", t.toh, t.tow, pwp, patch_read_conflicts(t.toh, t.tow, pwp, stride));
    }
    return 0;
}
