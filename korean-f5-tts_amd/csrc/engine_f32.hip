// libf5hip.so -- the engine instantiated for operand type float (see engine_impl.h).
#include "engine_impl.h"

template struct EngineOps<float>;
