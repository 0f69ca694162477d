/* Definitions behind oracle/shim/TH*.h: a "tensor" is {data, size[3]}. */
#include <stddef.h>
#include "TH/TH.h"
#include "THC/THC.h"
typedef struct { void *data; long size[3]; } stub_tensor;
THCState *state = NULL;
float *THFloatTensor_data(THFloatTensor *t) { return (float *)((stub_tensor *)t)->data; }
int *THIntTensor_data(THIntTensor *t) { return (int *)((stub_tensor *)t)->data; }
long THCudaTensor_size(THCState *s, const void *t, int dim) { (void)s; return ((const stub_tensor *)t)->size[dim]; }
