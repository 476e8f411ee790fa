"""`chex` stand-in (dev-only): only the type aliases the reference annotates with."""
from typing import Any

Array = Any
ArrayTree = Any
Numeric = Any
Scalar = Any
Shape = Any
PRNGKey = Any
