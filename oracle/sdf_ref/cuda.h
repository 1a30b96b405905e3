// TEST INFRASTRUCTURE: empty stand-in for <cuda.h> (see ATen/ATen.h in this directory)
#pragma once
