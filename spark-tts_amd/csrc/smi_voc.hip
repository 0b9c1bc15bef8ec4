// placeholder until the vocoder kernels land (next commit)
#include "smi_common.h"
extern "C" {
int smi_voc_arena_count(const smi_voc_cfg*) { return 0; }
int smi_voc_arena_entry(const smi_voc_cfg*, int, char*, int, size_t*, size_t*, int32_t*) { smi_set_error("vocoder not built"); return SMI_EINVAL; }
size_t smi_voc_arena_bytes(const smi_voc_cfg*) { return 0; }
int smi_voc_create(const smi_voc_cfg*, const void*, size_t, smi_voc**) { smi_set_error("vocoder not built"); return SMI_EINVAL; }
int smi_voc_destroy(smi_voc*) { return SMI_OK; }
int smi_voc_forward(smi_voc*, const int64_t*, const int32_t*, const int32_t*, int, int, float*, void*) { smi_set_error("vocoder not built"); return SMI_EINVAL; }
int smi_voc_debug_stage(smi_voc*, int, float*, size_t, size_t*, void*) { smi_set_error("vocoder not built"); return SMI_EINVAL; }
int smi_voc_num_launches(smi_voc*) { return 0; }
int smi_voc_time_launch(smi_voc*, int, int, float*, double*, char*, int, void*) { smi_set_error("vocoder not built"); return SMI_EINVAL; }
}
