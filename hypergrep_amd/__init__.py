"""MI355X-native multi-pattern line scanner with the hypergrep Python API (drop-in for `import hypergrep`)."""

from hypergrep_amd.utils import (  # the reference package re-exports exactly these names
    CALLBACK_TYPE,
    HS_FLAG_CASELESS,
    HS_FLAG_DOTALL,
    HS_FLAG_MULTILINE,
    HS_FLAG_SINGLEMATCH,
    RC_INVALID_FILE,
    Result,
    check_compatibility,
    configure_libraries,
    grep,
    prepare_patterns,
    scan,
)

__all__ = [
    "CALLBACK_TYPE", "HS_FLAG_CASELESS", "HS_FLAG_DOTALL", "HS_FLAG_MULTILINE", "HS_FLAG_SINGLEMATCH", "RC_INVALID_FILE", "Result",
    "check_compatibility", "configure_libraries", "grep", "prepare_patterns", "scan",
]
__version__ = "0.1.0"
