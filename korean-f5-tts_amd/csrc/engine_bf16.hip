// libf5hip.so -- the engine instantiated for operand type bf16_t (see engine_impl.h).
#include "engine_impl.h"

template struct EngineOps<bf16_t>;
