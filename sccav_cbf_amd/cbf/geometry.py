"""Orientation holder for bounding boxes.

Only the yaw of a box's rotation enters the barrier arithmetic (reference
cbf/obstacles.py:301, 330); the quaternion / 4x4 transform plumbing of the
reference's cbf/geometry.py serves the CARLA drivers and is not reproduced.
"""


class Rotation:
    def __init__(self, roll=0.0, pitch=0.0, yaw=0.0):
        self.roll = roll
        self.pitch = pitch
        self.yaw = yaw

    def __repr__(self):
        return "Rotation(roll=%r, pitch=%r, yaw=%r)" % (self.roll, self.pitch, self.yaw)
