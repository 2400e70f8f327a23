#include "../../lumen_min.h"
