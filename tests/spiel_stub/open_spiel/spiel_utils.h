#pragma once
#include "open_spiel/spiel.h"
