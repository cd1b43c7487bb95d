from .sequence import *
from .embedding import *
from .mlp import *
