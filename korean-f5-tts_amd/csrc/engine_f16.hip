// libf5hip.so -- the engine instantiated for operand type f16_t (see engine_impl.h).
#include "engine_impl.h"

template struct EngineOps<f16_t>;
