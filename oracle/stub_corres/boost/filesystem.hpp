// Forwarder (CorresApp.cpp:6); test infrastructure, see er_corres_stub.h.
#pragma once
#include "er_corres_stub.h"
