"""Robot-side (simulator) joint / body order of the Unitree G1 29-DoF asset.

Isaac Lab resolves these orders from the USD at run time (robot.data.joint_names / body_names); without the
simulator they have to be data.  Source: the order recorded by the reference at
motions/test/get_joint_name.py:231-232, identical to the ``dof_names`` / ``body_names`` stored in
``G1_dance.npz`` (which was recorded from the simulator, motions/record_data.py).  The env needs them to map
clip columns to robot columns (g1_amp_env.py:47-60).
"""

G1_JOINT_NAMES = [
    'left_hip_pitch_joint',
    'right_hip_pitch_joint',
    'waist_yaw_joint',
    'left_hip_roll_joint',
    'right_hip_roll_joint',
    'waist_roll_joint',
    'left_hip_yaw_joint',
    'right_hip_yaw_joint',
    'waist_pitch_joint',
    'left_knee_joint',
    'right_knee_joint',
    'left_shoulder_pitch_joint',
    'right_shoulder_pitch_joint',
    'left_ankle_pitch_joint',
    'right_ankle_pitch_joint',
    'left_shoulder_roll_joint',
    'right_shoulder_roll_joint',
    'left_ankle_roll_joint',
    'right_ankle_roll_joint',
    'left_shoulder_yaw_joint',
    'right_shoulder_yaw_joint',
    'left_elbow_joint',
    'right_elbow_joint',
    'left_wrist_roll_joint',
    'right_wrist_roll_joint',
    'left_wrist_pitch_joint',
    'right_wrist_pitch_joint',
    'left_wrist_yaw_joint',
    'right_wrist_yaw_joint',
]

G1_BODY_NAMES = [
    'pelvis',
    'imu_in_pelvis',
    'left_hip_pitch_link',
    'pelvis_contour_link',
    'right_hip_pitch_link',
    'waist_yaw_link',
    'left_hip_roll_link',
    'right_hip_roll_link',
    'waist_roll_link',
    'left_hip_yaw_link',
    'right_hip_yaw_link',
    'torso_link',
    'left_knee_link',
    'right_knee_link',
    'd435_link',
    'head_link',
    'imu_in_torso',
    'left_shoulder_pitch_link',
    'logo_link',
    'mid360_link',
    'right_shoulder_pitch_link',
    'left_ankle_pitch_link',
    'right_ankle_pitch_link',
    'left_shoulder_roll_link',
    'right_shoulder_roll_link',
    'left_ankle_roll_link',
    'right_ankle_roll_link',
    'left_shoulder_yaw_link',
    'right_shoulder_yaw_link',
    'left_elbow_link',
    'right_elbow_link',
    'left_wrist_roll_link',
    'right_wrist_roll_link',
    'left_wrist_pitch_link',
    'right_wrist_pitch_link',
    'left_wrist_yaw_link',
    'right_wrist_yaw_link',
    'left_rubber_hand',
    'right_rubber_hand',
]

# AMP key bodies, in observation order: right hand, left hand, right foot, left foot (g1_amp_env.py:40-45)
G1_KEY_BODY_NAMES = ["right_rubber_hand", "left_rubber_hand", "right_ankle_roll_link", "left_ankle_roll_link"]
# humanoid_28 (humanoid_amp_env.py:42); its robot-side joint order is not recorded anywhere in the reference
# (third-party asset HUMANOID_28_CFG): the clip order is used (unpinned)
HUMANOID_KEY_BODY_NAMES = ["right_hand", "left_hand", "right_foot", "left_foot"]

# The 23-DoF G1 (BASELINE.json configs[1]'s literal wording; the reference's assets and clips are all 29-DoF): the
# 29-DoF order above without the joints the 23-DoF model does not have -- waist roll / pitch and both wrists' pitch / yaw.
G1_23DOF_DROPPED = ('waist_roll_joint', 'waist_pitch_joint', 'left_wrist_pitch_joint', 'right_wrist_pitch_joint',
                    'left_wrist_yaw_joint', 'right_wrist_yaw_joint')
G1_23DOF_JOINT_NAMES = [n for n in G1_JOINT_NAMES if n not in G1_23DOF_DROPPED]
