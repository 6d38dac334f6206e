#pragma once
#include "ifeature_extractor.h"
