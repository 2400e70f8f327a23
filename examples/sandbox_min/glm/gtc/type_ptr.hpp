#include "../../glm_min.h"
