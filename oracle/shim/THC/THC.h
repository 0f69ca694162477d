/* stub of the THC symbols toolbox/nndistance/src/my_lib.c uses */
#pragma once
typedef struct THCState THCState;
typedef struct THCudaTensor THCudaTensor;
long THCudaTensor_size(THCState *state, const void *t, int dim);
