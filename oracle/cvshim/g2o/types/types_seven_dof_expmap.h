#include "se3quat.h"
