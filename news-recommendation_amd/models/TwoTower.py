"""TwoTowerModel: a news encoder + a user encoder + dot-product scoring (reference models/TwoTower.py:5-9)."""
from .BaseModel import TwoTowerBaseModel


class TwoTowerModel(TwoTowerBaseModel):
    def __init__(self, manager, newsEncoder, userEncoder):
        name = "-".join(("TwoTower", newsEncoder.name, userEncoder.name))
        super().__init__(manager, name=name)
        self.newsEncoder = newsEncoder
        self.userEncoder = userEncoder
