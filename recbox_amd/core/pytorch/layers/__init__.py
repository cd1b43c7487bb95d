from .sequence import *
from .embedding import *
