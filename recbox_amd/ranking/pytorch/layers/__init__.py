from .pooling import *
from .embeddings import *
from .interactions import *
from .blocks import *
from .attentions import *
