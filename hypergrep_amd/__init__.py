"""MI355X-native multi-pattern line scanner with the hypergrep Python API (drop-in for `import hypergrep`)."""

from hypergrep_amd.utils import CALLBACK_TYPE
from hypergrep_amd.utils import HS_FLAG_CASELESS
from hypergrep_amd.utils import HS_FLAG_DOTALL
from hypergrep_amd.utils import HS_FLAG_MULTILINE
from hypergrep_amd.utils import HS_FLAG_SINGLEMATCH
from hypergrep_amd.utils import RC_INVALID_FILE
from hypergrep_amd.utils import Result
from hypergrep_amd.utils import check_compatibility
from hypergrep_amd.utils import configure_libraries
from hypergrep_amd.utils import grep
from hypergrep_amd.utils import prepare_patterns
from hypergrep_amd.utils import scan

__version__ = "0.1.0"
