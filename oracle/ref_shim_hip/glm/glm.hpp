/* the GLM subset of oracle/ref_shim (one definition for the host and the device build of oracle/_ref) */
#include "../../ref_shim/glm/glm.hpp"
