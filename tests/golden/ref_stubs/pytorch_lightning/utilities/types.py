STEP_OUTPUT = object
