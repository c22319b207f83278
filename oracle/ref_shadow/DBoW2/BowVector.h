#include "FeatureVector.h"
