"""Drop-in for the reference's `arch` package (arch/__init__.py:1-3)."""
from .generators import define_Gen
from .discriminators import define_Dis
from .ops import set_grad, batch_groups
