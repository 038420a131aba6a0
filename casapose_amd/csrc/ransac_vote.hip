// placeholder -- implemented after the first end-to-end forward runs on the GPU
#include "common.h"
extern "C" size_t cp_ransac_workspace_bytes(int, int, int, int, int, int) { return 0; }
extern "C" int cp_ransac_vote_f32(const uint8_t*, const float*, int, int, int, int, int, int, int, const int32_t*, int, float,
                                  float, int, int, int, void*, float*, int32_t*, void*) {
    cp::set_error("cp_ransac_vote_f32: not implemented yet");
    return CP_ERR_INVALID;
}
