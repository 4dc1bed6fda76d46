from . import cartpole
from . import dummy
from . import synthetic
