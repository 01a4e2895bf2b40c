// Build shim (OURS): empty stand-in (nothing of it is used by the code compiled against this shim).
#pragma once
