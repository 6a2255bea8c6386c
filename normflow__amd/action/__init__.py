from .scalar_action import ScalarPhi4Action
