#include "common.h"
#include "kernels.h"
extern "C" const char* dist_strerror(int code) {
    switch (code) {
        case DIST_OK: return "ok";
        case DIST_ERR_ARG: return "invalid argument or unsupported shape";
        case DIST_ERR_STATE: return "invalid call order";
        case DIST_ERR_WORKSPACE: return "workspace too small or not bound";
        case DIST_ERR_UNBOUND: return "weights/buffers not bound";
        default: return code <= -1000 ? hipGetErrorString((hipError_t)(-code - 1000)) : "unknown error";
    }
}
extern "C" int dist_abi_version(void) { return 1; }
// ---- temporary stubs (replaced by the real engine) ----
struct dist_handle { int x; };
extern "C" {
int dist_create(const dist_config*, dist_handle**) { return DIST_ERR_STATE; }
void dist_destroy(dist_handle*) {}
const char* dist_last_error(const dist_handle*) { return ""; }
int dist_param_count(const dist_handle*, int) { return 0; }
const char* dist_param_name(const dist_handle*, int, int) { return ""; }
int dist_param_ndim(const dist_handle*, int, int) { return 0; }
int64_t dist_param_dim(const dist_handle*, int, int, int) { return 0; }
int64_t dist_param_offset(const dist_handle*, int, int) { return 0; }
int64_t dist_param_total(const dist_handle*, int) { return 0; }
int dist_param_group(const dist_handle*, int) { return 0; }
size_t dist_workspace_bytes(const dist_handle*) { return 0; }
size_t dist_packed_bytes(const dist_handle*) { return 0; }
int dist_bind(dist_handle*, float*, float*, const float*, float*, float*, void*, void*) { return DIST_ERR_STATE; }
int dist_pack_weights(dist_handle*, int, void*) { return DIST_ERR_STATE; }
int dist_vit_forward(dist_handle*, const float*, int, void*) { return DIST_ERR_STATE; }
int dist_branch_forward(dist_handle*, const float*, int, float*, float*, void*) { return DIST_ERR_STATE; }
int dist_branch_backward(dist_handle*, const float*, int, int, void*) { return DIST_ERR_STATE; }
int dist_loss(dist_handle*, const float*, int, float*, float*, void*) { return DIST_ERR_STATE; }
int dist_debug_tensor(dist_handle*, const char*, const void**, int64_t*, int*) { return DIST_ERR_STATE; }
}
