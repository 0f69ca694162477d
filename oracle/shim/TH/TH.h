/* stub of the two TH types/accessors toolbox/nndistance/src/my_lib.c uses */
#pragma once
typedef struct THFloatTensor THFloatTensor;
typedef struct THIntTensor THIntTensor;
float *THFloatTensor_data(THFloatTensor *t);
int *THIntTensor_data(THIntTensor *t);
