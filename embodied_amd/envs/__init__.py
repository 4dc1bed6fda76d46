from . import dummy
from . import synthetic
