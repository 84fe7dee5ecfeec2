/* vox_safetensors.h — read-only safetensors index over an mmap'd file.
 * (Role of the reference's voxtral_safetensors.c:204-429; weight bytes are handed to
 * the GPU straight from the mapping, nothing is converted or copied on the host.) */
#ifndef VOX_SAFETENSORS_H
#define VOX_SAFETENSORS_H
#include <stddef.h>
#include <stdint.h>

typedef enum { VOX_ST_BF16 = 0, VOX_ST_F16, VOX_ST_F32, VOX_ST_OTHER } vox_st_dtype_t;

typedef struct {
    char *name;
    vox_st_dtype_t dtype;
    int ndim;
    int64_t shape[8];
    const uint8_t *data;      /* pointer into the mapping */
    size_t nbytes;
} vox_st_tensor_t;

typedef struct {
    void *map;
    size_t map_size;
    int n_tensors;
    vox_st_tensor_t *tensors;   /* sorted by name for bsearch */
} vox_st_file_t;

vox_st_file_t *vox_st_open(const char *path);
void vox_st_close(vox_st_file_t *f);
const vox_st_tensor_t *vox_st_find(const vox_st_file_t *f, const char *name);
int64_t vox_st_numel(const vox_st_tensor_t *t);
/* malloc'd f32 copy of a BF16/F16/F32 tensor (small vectors only). */
float *vox_st_to_f32(const vox_st_tensor_t *t);

#endif
