/* cub -> hipCUB (the two device primitives the reference calls keep their signatures) */
#include "../cuda_on_hip.h"
#include <hipcub/hipcub.hpp>
namespace cub = hipcub;
