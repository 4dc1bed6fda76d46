from .train import train
from .eval_only import eval_only
from .pretrain import pretrain
from .train_eval import train_eval
from .actor_learner import actor_learner
