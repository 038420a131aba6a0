// placeholder -- implemented after the first end-to-end forward runs on the GPU
#include "common.h"
extern "C" size_t cp_ccl_workspace_bytes(int batch, int h, int w, int objects) { return (size_t)batch * h * w * sizeof(int) * 2; }
extern "C" int cp_ccl_filter_labels(const uint8_t*, int, int, int, int, int, void*, uint8_t*, void*) {
    cp::set_error("cp_ccl_filter_labels: not implemented yet");
    return CP_ERR_INVALID;
}
