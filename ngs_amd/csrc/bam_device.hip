// bam_device.hip -- BAM record parse on the device (second half).
#include <hip/hip_runtime.h>

#include "ingest_kernels.h"

namespace ngsq {} // namespace ngsq
